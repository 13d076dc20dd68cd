"""CPU, world sizes 2, 4 and 8 (gloo): the N > 1 path of bench.py — per-rank row shards, one equal-sized all_gather per frame or per
bucket of frames (RGBA16F, and the RGBA8 display buckets of vpt_renderer_play_into_display at 16 frames per collective), reassembly by
index_select — reproduces the full frame on every rank."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from vpt_amd.tiles import FrameGather, row_owner, local_rows
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    W, H = 37, 70
    for rows in (8, 5):
        g = FrameGather(dist, torch, W, H, torch.device("cpu"), rows_per_block=rows)
        assert g.rows == local_rows(H, world, rows) and g.shard() == (rank, world, rows)
        owner, lrow = row_owner(H, world, rows)
        # the "render": every pixel holds (global row, column, frame id, 1) — what a bit-exact shard would write
        for frame in range(3):
            b = frame & 1
            g.wait(b)
            g.send[b].zero_()
            for j in range(H):
                if owner[j] == rank:
                    g.send[b][lrow[j], :, 0] = j
                    g.send[b][lrow[j], :, 1] = torch.arange(W, dtype=torch.float16)
                    g.send[b][lrow[j], :, 2] = frame
                    g.send[b][lrow[j], :, 3] = 1
            g.gather(b)
            img = g.frame(b)
            assert img.shape == (H, W, 4)
            assert (img[:, :, 0] == torch.arange(H, dtype=torch.float16)[:, None]).all()
            assert (img[:, :, 1] == torch.arange(W, dtype=torch.float16)[None, :]).all()
            assert (img[:, :, 2] == frame).all() and (img[:, :, 3] == 1).all()
    # bucketed exchange: F frames per all_gather through the streaming interface (what bench.py drives), with a partial
    # bucket flushed by last_frame() and a new bucket started after it
    for F in (3, 4):
        g = FrameGather(dist, torch, W, H, torch.device("cpu"), frames_per_gather=F)
        owner, lrow = row_owner(H, world, 8)
        mine = torch.as_tensor(np.nonzero(owner == rank)[0])
        for frame in range(11):
            t = g.acquire()
            assert t.shape == (g.rows, W, 4)
            t.zero_()
            t[torch.as_tensor(lrow)[mine], :, 0] = mine.to(torch.float16)[:, None]
            t[torch.as_tensor(lrow)[mine], :, 2] = frame
            g.commit()
            if frame in (0, 2, 3, 6, 10):                     # mid-bucket, bucket end, after a flush
                img = g.last_frame()
                assert img.shape == (H, W, 4)
                assert (img[:, :, 0] == torch.arange(H, dtype=torch.float16)[:, None]).all(), (F, frame)
                assert (img[:, :, 2] == frame).all(), (F, frame)
                assert (g.last_sent()[torch.as_tensor(lrow)[mine], :, 2] == frame).all()
        g.flush(); g.wait_all()
    # whole buckets at once (acquire_bucket / commit_bucket: what bench.py pairs with vpt_renderer_play_into), mixed with single frames
    F = 4
    g = FrameGather(dist, torch, W, H, torch.device("cpu"), frames_per_gather=F)
    owner, lrow = row_owner(H, world, 8)
    mine = torch.as_tensor(np.nonzero(owner == rank)[0])
    rows_mine = torch.as_tensor(lrow)[mine]
    frame = 0
    for step in range(5):
        bucket = g.acquire_bucket()
        assert bucket is not None and bucket.shape == (F, g.rows, W, 4)
        bucket.zero_()
        for j in range(F):
            bucket[j][rows_mine, :, 0] = mine.to(torch.float16)[:, None]
            bucket[j][rows_mine, :, 2] = frame + j
        g.commit_bucket()
        frame += F
        img = g.last_frame()
        assert (img[:, :, 0] == torch.arange(H, dtype=torch.float16)[:, None]).all() and (img[:, :, 2] == frame - 1).all(), step
        if step == 2:                                         # two single frames open a bucket: no whole bucket until it is flushed
            for _ in range(2):
                t = g.acquire(); t.zero_(); t[rows_mine, :, 2] = frame; g.commit(); frame += 1
            assert g.acquire_bucket() is None
            assert (g.last_frame()[:, :, 2] == frame - 1).all()     # (flushes the partial bucket)
    g.flush(); g.wait_all()
    # the display form (bench.py's config.display_gather_form): RGBA8 texels, 16 frames per all_gather — whole buckets, then a partial one
    F = 16
    g = FrameGather(dist, torch, W, H, torch.device("cpu"), frames_per_gather=F, texel='rgba8')
    assert g.texel_bytes == 4 and g._send[0].dtype == torch.uint8
    frame = 0
    rows8 = (mine %% 251).to(torch.uint8)[:, None]
    for step in range(3):
        bucket = g.acquire_bucket()
        assert bucket is not None and bucket.shape == (F, g.rows, W, 4) and bucket.dtype == torch.uint8
        bucket.zero_()
        for j in range(F):
            bucket[j][rows_mine, :, 0] = rows8
            bucket[j][rows_mine, :, 1] = torch.arange(W, dtype=torch.uint8)
            bucket[j][rows_mine, :, 2] = (frame + j) %% 256
            bucket[j][rows_mine, :, 3] = 255
        g.commit_bucket()
        frame += F
        img = g.last_frame()
        assert img.shape == (H, W, 4) and img.dtype == torch.uint8
        assert (img[:, :, 0] == (torch.arange(H) %% 251).to(torch.uint8)[:, None]).all(), step
        assert (img[:, :, 1] == torch.arange(W, dtype=torch.uint8)[None, :]).all() and (img[:, :, 2] == (frame - 1) %% 256).all() and (img[:, :, 3] == 255).all(), step
    for k in range(5):                                        # a partial bucket of 5 frames, flushed by last_frame()
        t = g.acquire(); t.zero_(); t[rows_mine, :, 0] = rows8; t[rows_mine, :, 2] = (frame %% 256); g.commit(); frame += 1
    img = g.last_frame()
    assert (img[:, :, 0] == (torch.arange(H) %% 251).to(torch.uint8)[:, None]).all() and (img[:, :, 2] == (frame - 1) %% 256).all()
    g.flush(); g.wait_all()
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


import pytest


@pytest.mark.parametrize("world", [2, 4, 8])
def test_frame_gather_gloo(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o
