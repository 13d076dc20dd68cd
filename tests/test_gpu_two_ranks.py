"""GPU (needs TWO GPUs in the box; skipped on the one-GPU boxes of the pool): the native RCCL gather pipeline with a real
two-rank communicator — the rooted grouped ncclSend / ncclRecv path, the in-place render into the root's receive slot, the
all_gather path and the ring's reverse edge, none of which a one-rank communicator executes (vpt_hip.hip gather_enqueue_frame).
Two fresh child processes are started before anything in them touches a GPU (tests/two_rank_worker.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpu_count():
    import torch
    return torch.cuda.device_count()                     # does not initialise the GPU


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs (the pool's boxes have one): run on the multi-GPU node")
def test_native_gather_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("two-rank gather did not finish in 300 s (a hung collective?)")
        outs.append(out.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_torch_pipeline_on_one_rank_frame_by_frame_and_with_bucket_kernels():
    """bench.py's N > 1 code path on a one-rank RCCL group (--force-dist 1 --gather torch), small sizes: buckets of 16 frames through
    vpt_renderer_play_into + one all_gather each, the gathered frame bit-compared with the same frames rendered unsharded — frame by frame
    (the line), with VPT_OPTION_BUCKET_KERNEL (config.bucket_kernel_form: one launch per tile class and bucket) and gathering the frames as the
    tone mapper shows them (config.display_gather_form: RGBA8 buckets, checked against render() + toneMapper.render() unsharded).  A fresh process:
    a GPU-initialised torch stays out of this one."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "1", "--gather", "torch", "--volume", "64", "--width", "640",
                        "--height", "368", "--steps", "40", "--warmup", "8", "--repeats", "2", "--other-configs", "0", "--cpu-baseline", "0",
                        "--stream-probe", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["frame_check"] is True
    assert line["config"]["frames_per_gather"] == 16 and line["config"]["bucket_kernel"] is False
    b = line["config"]["bucket_kernel_form"]
    assert b["frame_check"] is True and b["frames_per_launch"] == 16
    assert b["bucket_launches"] >= (1000 + 2 * 32) // 16         # the warm-up and both timed blocks went through the bucket kernels
    d = line["config"]["display_gather_form"]                    # ... and gathering the tone-mapped RGBA8 frames (vpt_renderer_play_into_display)
    assert d["frame_check"] is True and d["bucket_launches"] >= (1000 + 2 * 32) // 16
    assert d["bytes_per_frame_and_xgmi_link"] == 640 * 368 * 4
    assert line["config"]["display_cadence_form"]["frame_check"] is True


@pytest.mark.parametrize("world", [2, 3])
def test_bench_rehearsal_with_several_ranks_on_one_gpu(world):
    """bench.py --gpus N --rehearsal 1: N rank processes on the box's ONE GPU with the collectives over gloo (RCCL refuses two ranks on one
    device).  Everything of the N > 1 path above the collective's transport runs with a real world size: every rank's shard (interleaved row
    blocks, padding: 368 rows over 3 ranks), buckets of 16 frames through vpt_renderer_play_into, the gathered frame bit-compared with the same
    frames unsharded on every rank, the bucket-kernel and RGBA8 display forms, the single-GPU reference, max-over-ranks timing, ONE JSON line
    from rank 0.  Its figures measure nothing (the ranks share one GPU); the line says so."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--rehearsal", "1", "--volume", "64",
                                       "--width", "640", "--height", "368", "--steps", "40", "--warmup", "8", "--repeats", "2", "--stream-probe", "0",
                                       "--warmup-seconds", "0.05", "--watchdog", "500"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=560))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("the rehearsal did not finish in 560 s")
    assert all(p.returncode == 0 for p in procs), "\n".join(e.decode(errors="replace")[-2000:] for _, e in outs)
    lines = [o.decode().strip() for o, _ in outs]
    assert all(l == "" for l in lines[1:]), "only rank 0 prints"
    line = json.loads(lines[0].splitlines()[-1])
    assert line["n_gpus"] == world and "rehearsal" in line and line["frame_check"] is True
    assert line["config"]["frames_per_gather"] == 16
    assert line["config"]["bucket_kernel_form"]["frame_check"] is True
    assert line["config"]["display_gather_form"]["frame_check"] is True
    c = line["config"]["display_cadence_form"]                      # 16 passes per launch on every rank, the 16th frame gathered
    assert c["frame_check"] is True and c["passes_per_shown_frame"] == 16
    assert line["config"]["single_gpu_reference"]["frame_by_frame_ms"] > 0
    assert line["config"]["speedup_like_for_like"]["display_cadence_form_vs_single_gpu_same_form"] > 0
    assert 640 * 368 * 8 <= line["config"]["samples_per_step"] <= 640 * (368 + 8 * world) * 8      # every pixel of the frame (+ at most one padding block per rank)
