#!/usr/bin/env python3
"""Host cost of the torch.distributed gather pipeline (the N > 1 default of bench.py) on a one-rank RCCL group, with rank 3 of 8's share of
the headline frame (1920 x 136 rows) and whole frames, bucket by bucket as bench.py drives it (acquire_bucket -> vpt_renderer_play_into ->
join -> all_gather of the bucket): where do the microseconds of the host loop go?  VPT_PROBE_F = frames per bucket (default 8),
VPT_PROBE_BUCKET = 1: VPT_OPTION_BUCKET_KERNEL (one launch per tile class and bucket); VPT_PROBE_DISPLAY = 1: RGBA8 buckets of the frames as the
Artistic tone mapper shows them (vpt_renderer_play_into_display)."""
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
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        for H in (136, 1080):
            W, F = 1920, int(os.environ.get("VPT_PROBE_F", "8"))
            bucket_kernel = int(os.environ.get("VPT_PROBE_BUCKET", "0"))
            display = int(os.environ.get("VPT_PROBE_DISPLAY", "0"))
            ctx = vpt_amd.Context(0, stream=stream.cuda_stream)
            gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(512, noise=48.0), 'linear')
            # H = 136: the rows rank 3 of 8 renders of the 1080-row frame (interleaved 8-row blocks); the one-rank gather moves frames of that size
            shard, full_h = ((3, 8, 8), 1080) if H == 136 else ((0, 1, 8), H)
            r = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / full_h), None, {'resolution': (W, full_h), 'transform': Transform(Node()), 'rng': GoldenRatioRng(), 'shard': shard})
            assert int(r.local_rows()) == H
            r.set_option(N.OPTION_FAST_MATH, 1); r.set_option(N.OPTION_SPLIT_STREAMS, 2);
            r.set_option(N.OPTION_BUCKET_KERNEL, bucket_kernel)
            r.reset()
            gather = FrameGather(dist, torch, W, H, device, always_collective=True, frames_per_gather=F, texel='rgba8' if display else 'rgba16f')
            nbytes = gather.send[0].numel() * (1 if display else 2)
            tm = None
            if display:
                tm = vpt_amd.ToneMapperFactory('artistic')(ctx, r, {'resolution': (W, full_h)})
                tm.set_option(N.TONEMAPPER_OPTION_TABLE, N.TONEMAPPER_TABLE_ALWAYS)
                r.render(); tm.render()
            acc = {"acquire": 0.0, "play_into": 0.0, "join": 0.0, "commit": 0.0}
            n = 300

            def bucket(timing):
                t0 = time.perf_counter(); b = gather.acquire_bucket()
                t1 = time.perf_counter()
                if tm is not None:
                    r.play_into_display(tm, F, b.data_ptr(), nbytes)
                else:
                    r.play_into(F, b.data_ptr(), nbytes)
                t2 = time.perf_counter(); r.join()
                t3 = time.perf_counter(); gather.commit_bucket()
                t4 = time.perf_counter()
                if timing:
                    acc["acquire"] += t1 - t0; acc["play_into"] += t2 - t1; acc["join"] += t3 - t2; acc["commit"] += t4 - t3
            for k in range(50):
                bucket(False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(n):
                bucket(True)
            host = time.perf_counter() - t0
            gather.wait_all(); torch.cuda.synchronize()
            total = time.perf_counter() - t0
            out["H%d" % H] = {"frames_per_bucket": F, "bucket_kernel": bool(bucket_kernel), "display_rgba8": bool(display), "bucket_launches": r.bucket_launches(), "us_per_frame_total": total / (n * F) * 1e6, "us_per_frame_host_loop": host / (n * F) * 1e6,
                              **{"host_us_per_bucket_" + k: v / n * 1e6 for k, v in acc.items()}}
            if tm is not None:
                tm.destroy()
            r.set_render_target(0, 0); r.destroy(); gvol.destroy(); ctx.destroy()
    print(json.dumps(out, indent=1))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
