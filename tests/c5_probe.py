#!/usr/bin/env python3
"""C5-sized probe on ONE GPU (MCM 2048^3 u8 @ 3840x2160, sharded over 8 GPUs).

The 8-GPU run itself belongs to the driver; this measures and checks what one rank of it does:
  * the 2048^3 volume goes through the 64-bit brick-offset tables (16 GiB of bricks > 4 GiB),
  * rank `--rank` of 8 renders its interleaved row blocks: kernel time per frame,
  * a band of its rows is compared bit for bit with the CPU oracle run on the same 2048^3 volume,
  * the full 3840x2160 frame on one GPU, for the strong-scaling denominator.
Writes one JSON object (stdout, and --out).  A test script (run by hand: python tests/c5_probe.py), kept out of the
pytest collection because it needs ~10 GiB of host memory and a minute of volume generation; uses oracle/ as the checker.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repository root
sys.path.insert(0, ROOT)


def _slab(args):
    from vpt_amd.synthetic import sphere_volume
    n, z0, z1 = args
    return z0, sphere_volume(n, noise=48.0, z_range=(z0, z1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", type=int, default=2048)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--workers", type=int, default=14)
    ap.add_argument("--oracle-rows", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()

    import vpt_amd
    from vpt_amd import _native as N
    from vpt_amd.scene import Node, Transform, default_camera, mvp_inverse_matrix
    from vpt_amd.synthetic import GoldenRatioRng
    from oracle import oracle

    n, W, H = a.volume, a.width, a.height
    t0 = time.time()
    vol = np.empty((n, n, n), dtype=np.uint8)
    step = 32
    with mp.get_context("fork").Pool(a.workers) as pool:            # forked before anything touches the GPU
        for z0, s in pool.imap_unordered(_slab, [(n, z, min(z + step, n)) for z in range(0, n, step)]):
            vol[z0:z0 + s.shape[0]] = s
    t_gen = time.time() - t0
    print("volume %d^3 generated in %.1f s" % (n, t_gen), file=sys.stderr, flush=True)

    ctx = vpt_amd.Context(0)
    t0 = time.time()
    gvol = vpt_amd.Volume.from_array(ctx, vol, 'linear')
    ctx.synchronize()
    t_up = time.time() - t0
    bricked = gvol.bricked_bytes()
    print("uploaded + bricked in %.1f s (%d bytes of bricks)" % (t_up, bricked), file=sys.stderr, flush=True)

    camera = default_camera(W / H)
    transform = Transform(Node())
    m = mvp_inverse_matrix(camera, transform)

    def renderer(**opts):
        o = {'resolution': (W, H), 'transform': transform, 'rng': GoldenRatioRng()}
        o.update(opts)
        return vpt_amd.MCMRenderer(ctx, gvol, camera, None, o)

    out = {"volume": n, "width": W, "height": H, "bricked_bytes": bricked, "wide_tables": bricked > (4 << 30),
           "generate_s": t_gen, "upload_brickify_s": t_up}

    # ---- parity: two passes of the shard, a band of its rows against the oracle on the same volume
    passes = 2
    r = renderer(shard=(a.rank, a.world, 8))
    r.reset()
    for _ in range(passes):
        r.render()
    rad = r.read(N.BUFFER_MCM_RADIANCE)
    rows = r.global_rows()
    assert r.sample_count() == int((rows >= 0).sum()) * W * 8 * passes
    # the row block of this rank nearest the middle of the image
    mid = int(np.argmin(np.abs(rows - H // 2)))
    l0 = (mid // 8) * 8
    l1 = l0 + a.oracle_rows
    y0, y1 = int(rows[l0]), int(rows[l1 - 1]) + 1
    assert y1 - y0 == a.oracle_rows, "band must lie inside one row block"
    osc = oracle.OracleScene(vol, 'linear')
    o = oracle.OracleRenderer('mcm', osc, W, H)
    rng = GoldenRatioRng()
    nthreads = len(os.sched_getaffinity(0))
    fr = oracle.make_frame(W, H, m, seed=np.float32(rng()), y0=y0, y1=y1, nthreads=nthreads)
    o.reset(fr)
    for _ in range(passes):
        fr.seed = float(np.float32(rng()))
        o.integrate(fr)
    want = o.state[3].reshape(H, W, 4)[y0:y1]
    same = bool((rad[l0:l1].view(np.uint32) == want.view(np.uint32)).all())
    out["oracle_band"] = {"rows": [y0, y1], "bit_identical": same, "pixels": (y1 - y0) * W}
    print("oracle band rows %d..%d bit-identical: %s" % (y0, y1, same), file=sys.stderr, flush=True)

    # ---- timing: the rank's share of C5
    def timed(r, frames):
        r.reset()
        for _ in range(10):
            r.render()
        ctx.synchronize()
        r.clear_sample_count(); r.set_profiling(4)
        t0 = time.perf_counter()
        for _ in range(frames):
            r.render()
        t_enq = time.perf_counter() - t0
        ctx.synchronize()
        dt = time.perf_counter() - t0
        ms, launches = r.profile()
        r.set_profiling(False)
        return {"ms_per_frame": dt / frames * 1e3, "enqueue_ms_per_frame": t_enq / frames * 1e3, "kernel_avg_ms": ms / max(launches, 1), "samples_per_s": r.sample_count() / dt}

    out["shard_1_of_%d_first_run" % a.world] = timed(r, a.frames)
    out["shard_1_of_%d" % a.world] = timed(r, a.frames)
    r.destroy()
    full = renderer()
    out["full_frame_one_gpu"] = timed(full, a.frames)
    # sharded rows == the same rows of the full frame
    full.rng = GoldenRatioRng(); full.reset()
    sh = renderer(shard=(a.rank, a.world, 8)); sh.reset()
    for _ in range(passes):
        full.render(); sh.render()
    fimg, simg, rows = full.getTexture(), sh.getTexture(), sh.global_rows()
    out["shard_equals_full_rows"] = bool((fimg[rows[rows >= 0]].view(np.uint16) == simg[rows >= 0].view(np.uint16)).all())
    full.destroy(); sh.destroy(); gvol.destroy(); ctx.destroy()
    s = json.dumps(out)
    print(s)
    if a.out:
        with open(a.out, "w") as f:
            f.write(s + "\n")
    if not (same and out["shard_equals_full_rows"]):
        raise SystemExit("C5 probe: parity failed")


if __name__ == "__main__":
    main()
