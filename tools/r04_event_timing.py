#!/usr/bin/env python3
"""Where a wave of the HIT-tile kernel (k_mcm_integrate on the tiles the cube projects onto) spends its life, measured by an
INSTRUMENTED build of the library: make -C vpt_amd/csrc OUT=../../gpurun_ab/timing.so EXTRA=-DVPT_EVENT_TIMING (vpt_kernels_mcm.h:
a 100 MHz wave clock read at wave-uniform points of the event loop, each read behind the s_waitcnt of what the phase produced).
Per configuration: nanoseconds per event and wave of the five phases of an event, the wave's prologue and epilogue, the sum against
the kernel's own duration (HIP events), with the HIT kernel alone on the chip (one stream: HIT then MISS) and beside the MISS-tile
kernel (two streams).  Writes gpurun_out/r04/hit_kernel_latency.json (copied to profiles/r04_hit_kernel_latency.json)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="gpurun_ab/timing.so")
ap.add_argument("--out", default="gpurun_out/r04/hit_kernel_latency.json")
ap.add_argument("--volumes", default="512,1024")
ap.add_argument("--frames", type=int, default=200)
args = ap.parse_args()
os.environ["VPT_HIP_LIBRARY"] = os.path.abspath(args.lib)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vpt_amd
from vpt_amd import _native as N
from vpt_amd.scene import default_camera, Transform, Node
from vpt_amd.synthetic import sphere_volume, GoldenRatioRng

PHASES = ["free_path_pcg_log_move_bounds", "cell_lds_tables_loads_issued", "load_flight_atlas_or_brick", "blend_and_transfer_function_lds",
          "wheel_probabilities_path_end_or_scatter"]
W, H = 1920, 1080


def volume(n):
    from concurrent.futures import ThreadPoolExecutor
    v = np.empty((n, n, n), dtype=np.uint8)

    def slab(z0):
        v[z0:z0 + 16] = sphere_volume(n, noise=48.0, z_range=(z0, min(n, z0 + 16)))
    with ThreadPoolExecutor(max_workers=min(14, len(os.sched_getaffinity(0)))) as ex:
        list(ex.map(slab, range(0, n, 16)))
    return v


def timing(r):
    out = (C.c_uint64 * 17)()
    f = N.lib().vpt_probe_event_timing
    f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    f.restype = C.c_int
    N.check(f(r._h, out))
    return [int(x) for x in out]


ctx = vpt_amd.Context(0)
result = {"_what": __doc__.strip().split("\n\n")[0].replace("\n", " "), "_unit": "nanoseconds per wave (10 ns clock ticks x 10), means over every wave of %d frames" % args.frames,
          "_note": "the instrumented kernel serialises what the shipped kernel overlaps inside a wave (the marks wait for vmcnt / lgkmcnt): per-wave phase "
                   "times are an upper bound; `kernel_us_hip_events` of the instrumented and of the shipped library say by how much the whole kernel differs",
          "configs": []}
for n in [int(x) for x in args.volumes.split(",")]:
    vol = volume(n)
    gvol = vpt_amd.Volume.from_array(ctx, vol, 'linear')
    del vol
    for fast in (1, 0):
        for split in (1, 2):
            for records in (0, 1):
                r = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
                r.set_option(N.OPTION_FAST_MATH, fast)
                r.set_option(N.OPTION_SPLIT_STREAMS, split)
                r.set_option(N.OPTION_TILE_CLASSES, 2)                  # 2: the class kernels on one stream as well (HIT, then MISS: each alone on the chip)
                r.set_option(N.OPTION_COLUMN_RECORDS, records)
                r.reset()
                hit, miss, _ = r.tile_classes()
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.3:
                    for _ in range(50):
                        r.render()
                    ctx.synchronize()
                timing(r)                                               # clear
                r.set_profiling(1)
                ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.frames):
                    r.render()
                ctx.synchronize()
                wall = (time.perf_counter() - t0) / args.frames * 1e6
                ms, nl = r.profile(); ms2, nl2 = r.profile_side()
                r.set_profiling(False)
                t = timing(r)
                waves = max(t[8], 1)
                mark = 10.0 * t[7] / waves / 8.0                          # one mark's own cost (clock read + wait), per event
                per_event = [max(0.0, 10.0 * t[k] / waves / 8.0 - mark) for k in range(5)]
                entry = {"volume": n, "arithmetic": "fast-math" if fast else "bit-exact", "column_records": bool(records),
                         "hit_kernel": "alone on the chip (one stream: HIT, then MISS)" if split == 1 else "beside the MISS-tile kernel (two streams)",
                         "hit_tiles": hit, "miss_tiles": miss, "waves_per_frame": waves / args.frames,
                         "frame_us_wall": wall, "kernel_us_hip_events": {"hit": (ms / max(nl, 1) - ((ms2 / nl2) if (nl2 and split == 1) else 0.0)) * 1e3, "miss": (ms2 / nl2 * 1e3) if nl2 else None},
                         "per_event_ns": dict(zip(PHASES, per_event)), "event_ns": sum(per_event), "mark_ns_subtracted_from_every_phase": mark,
                         "prologue_ns_state_load_lds_staging_seed": 10.0 * t[5] / waves, "epilogue_ns_state_and_frame_stores": 10.0 * t[6] / waves,
                         "wave_lifetime_ns_instrumented": 10.0 * sum(t[:8]) / waves,
                         "last_launch_timeline_us_after_the_first_wave_started": {
                             "waves": t[9], "wave_start_p50_p90_max": [t[10] / 100.0, t[11] / 100.0, t[12] / 100.0],
                             "wave_end_p10_p50_p90_max": [t[13] / 100.0, t[14] / 100.0, t[15] / 100.0, t[16] / 100.0]}}
                result["configs"].append(entry)
                print(json.dumps(entry), flush=True)
                r.destroy()
    gvol.destroy()
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(result, open(args.out, "w"), indent=1)
ctx.destroy()
