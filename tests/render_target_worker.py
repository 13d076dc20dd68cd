#!/usr/bin/env python3
"""Worker of tests/test_gpu_parity.py::test_mcm_split_streams_with_a_caller_owned_render_target (a fresh process: torch's GPU
state stays out of the pytest process).  A frame rendered into caller memory (vpt_renderer_set_render_target) is read by work
the CALLER enqueues on the context's stream right behind render(): with VPT_OPTION_SPLIT_STREAMS on, such passes must not leave
rows on a side stream (vpt_renderer_play_into may, and joins before it returns).  The context runs on torch's current stream; the consumer is a torch copy on that stream, with no library
call in between.  Exit code 0 and a final line "OK" = every copied frame equals the one-stream run's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import vpt_amd
    from vpt_amd import _native as N
    from vpt_amd.scene import default_camera, Transform, Node
    from vpt_amd.synthetic import sphere_volume, GoldenRatioRng

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    W, H = 640, 480
    ctx = vpt_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(64, noise=40.0), 'linear')
    cam, tr = default_camera(W / H), Transform(Node())
    outs = []
    # one stream; three ranges asked for but kept on the context's stream by the library (a caller-owned target); vpt_renderer_play_into, whose
    # passes do run on every stream (here: the HIT | MISS kernels of the tile classes + a third stream) and are joined by the call itself
    for split, through_play_into in ((1, False), (3, False), (3, True)):
        r = vpt_amd.MCMRenderer(ctx, gvol, cam, None, {'resolution': (W, H), 'transform': tr, 'rng': GoldenRatioRng()})
        r.set_option(N.OPTION_SPLIT_STREAMS, split)
        target = torch.zeros((H, W, 4), dtype=torch.float16, device=dev)
        if not through_play_into:
            r.set_render_target(target.data_ptr(), target.numel() * 2)
        r.reset()
        copies = []
        for _ in range(6):
            if through_play_into:
                r.play_into(1, target.data_ptr(), target.numel() * 2)
            else:
                r.render()
            copies.append(target.clone())                 # enqueued on the context's stream, straight behind the pass: no library call in between
        torch.cuda.synchronize()
        outs.append([c.cpu().numpy() for c in copies])
        r.set_render_target(0, 0)
        r.destroy()
    ok = True
    for which in (1, 2):
        for k, (a, b) in enumerate(zip(outs[0], outs[which])):
            same = bool((a.view(np.uint16) == b.view(np.uint16)).all())
            if not same:
                sys.stderr.write("configuration %d, frame %d: the copy behind a 3-range pass differs from the one-stream run's\n" % (which, k))
            ok = ok and same
    last = outs[0][-1]
    ok = ok and bool(np.isfinite(last.astype(np.float32)).all()) and bool((last[..., 3] == 1).all())
    gvol.destroy(); ctx.destroy()
    print("OK" if ok else "FAILED")
    sys.stdout.flush()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
