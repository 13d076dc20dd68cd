#!/usr/bin/env python3
"""Worker of tests/test_gpu_two_ranks.py and of bench.py's native-gather preflight: one rank of an N-rank run of the NATIVE RCCL gather
pipeline (vpt_gather_*).
Started as a fresh process per rank (RANK / WORLD_SIZE / MASTER_* in the environment) before anything touches a GPU.
For root = 0 and root = -1 (all_gather): more than two turns of the buffer ring, the gathered frame of several frames compared
bit for bit with the same frames rendered UNSHARDED on this rank's GPU; the per-rank verdicts are all-reduced (MIN), so one bad
rank fails every rank.  Exit code 0 = all good."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    import vpt_amd
    from vpt_amd.scene import default_camera, Transform, Node
    from vpt_amd.synthetic import sphere_volume, GoldenRatioRng
    from vpt_amd.tiles import RcclFrameGather
    from vpt_amd import _native as N

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    gpu = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(gpu)
    device = torch.device("cuda", gpu)
    dist.init_process_group("nccl", device_id=device)
    W, H = 200, 136
    vol = sphere_volume(48, noise=40.0)
    ctx = vpt_amd.Context(gpu)
    gvol = vpt_amd.Volume.from_array(ctx, vol, 'linear')
    cam, tr = default_camera(W / H), Transform(Node())

    def renderer(shard=None):
        o = {'resolution': (W, H), 'transform': tr, 'rng': GoldenRatioRng()}
        if shard:
            o['shard'] = shard
        r = vpt_amd.MCMRenderer(ctx, gvol, cam, None, o)
        r.extinction = 4
        # the form bench.py runs: fast arithmetic; the sharded renderer with its pass on two streams (tile classes: HIT | MISS kernels)
        r.set_option(N.OPTION_FAST_MATH, 1)
        if shard:
            r.set_option(N.OPTION_SPLIT_STREAMS, 2)
        r.reset()
        return r

    ok = True
    ids = [RcclFrameGather.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    sharded = renderer((rank, world, 8))
    g = RcclFrameGather(sharded, ids[0], rank, world)
    whole = renderer()
    frames = 0
    for root in (0, -1, world - 1):
        g.set_root(root)
        for k in range(2 * 16 + 5):                      # VPT_GATHER_RING = 16: the ring wraps twice
            g.render(); whole.render(); frames += 1
            if k in (0, 15, 16, 33, 36) and g.receives():
                same = bool((g.frame().view(np.uint16) == whole.getTexture().view(np.uint16)).all())
                if not same:
                    sys.stderr.write("rank %d: root %d frame %d differs from the unsharded frame\n" % (rank, root, k))
                ok = ok and same
        g.synchronize()
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    g.destroy(); sharded.destroy(); whole.destroy(); gvol.destroy(); ctx.destroy()
    dist.destroy_process_group()
    print("rank %d: %s after %d frames" % (rank, "ok" if int(flag[0]) else "MISMATCH", frames))
    sys.exit(0 if int(flag[0]) else 1)


if __name__ == "__main__":
    main()
