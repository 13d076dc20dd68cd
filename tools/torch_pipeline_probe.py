#!/usr/bin/env python3
"""Host cost per frame of the torch.distributed gather pipeline (the N > 1 default of bench.py) on a one-rank RCCL group, with
frames of a 1/8 shard's size: where do the microseconds of the host loop go?"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch                                                       # noqa: E402
import torch.distributed as dist                                   # noqa: E402
import vpt_amd                                                     # noqa: E402
from vpt_amd import _native as N                                   # noqa: E402
from vpt_amd.scene import Node, Transform, default_camera          # noqa: E402
from vpt_amd.synthetic import GoldenRatioRng, sphere_volume        # noqa: E402
from vpt_amd.tiles import FrameGather                              # noqa: E402


def main():
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=device)
    out = {}
    for H in (136, 1080):
        W = 1920
        ctx = vpt_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(512, noise=48.0), 'linear')
        r = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng(), 'shard': (0, 1, 8)})
        r.set_option(N.OPTION_FAST_MATH, 1); r.reset()
        gather = FrameGather(dist, torch, W, H, device, always_collective=True)
        nbytes = gather.send[0].numel() * 2
        acc = {"wait": 0.0, "target": 0.0, "render": 0.0, "gather": 0.0}
        n = 600

        def frame(k, timing):
            b = k & 1
            t0 = time.perf_counter(); gather.wait(b)
            t1 = time.perf_counter(); r.set_render_target(gather.send[b].data_ptr(), nbytes)
            t2 = time.perf_counter(); r.render()
            t3 = time.perf_counter(); gather.gather(b)
            t4 = time.perf_counter()
            if timing:
                acc["wait"] += t1 - t0; acc["target"] += t2 - t1; acc["render"] += t3 - t2; acc["gather"] += t4 - t3
        for k in range(100):
            frame(k, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            frame(k, True)
        host = time.perf_counter() - t0
        gather.wait(0); gather.wait(1); torch.cuda.synchronize()
        total = time.perf_counter() - t0
        out["H%d" % H] = {"us_per_frame_total": total / n * 1e6, "us_per_frame_host_loop": host / n * 1e6,
                          **{"host_us_" + k: v / n * 1e6 for k, v in acc.items()}}
        r.set_render_target(0, 0); r.destroy(); gvol.destroy(); ctx.destroy()
    print(json.dumps(out, indent=1))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
