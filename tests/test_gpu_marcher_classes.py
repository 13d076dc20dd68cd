"""GPU: tile classes for the accumulating ray marchers (MIP, EAM, ISO, MCS, Depth; vpt_hip.hip marcher_track): after one whole fused
pass since the reset, and while every pass uses the reset's matrix, a fused render() launches only the tiles some ray of which
can meet the cube — the others hold final values (their accumulator sits at a fixed point of the integrate pass, their texels
of the render buffer do not change).  Everything readable must be identical with VPT_OPTION_TILE_CLASSES on and off, and equal
to the CPU oracle, through matrix changes without a reset, hook-by-hook passes in between, mixes other than 1/n, translucent
environment maps, shards, split streams and frame sequences."""
import numpy as np
import pytest

import vpt_amd
from vpt_amd import _native as N
from vpt_amd.synthetic import colour_tf, GoldenRatioRng

from conftest import orbit_camera
from test_gpu_parity import Scene, to_frame, assert_same_bits, env_map

pytestmark = pytest.mark.gpu
KINDS = ["mip", "eam", "iso", "mcs", "depth"]


def far_scene(gpu_ctx, oracle, w=208, h=144, env=None, cam=(0.7, -0.3, 3.2)):
    return Scene(gpu_ctx, oracle, 24, w, h, tf=colour_tf(48, 1), env=env, camera=orbit_camera(w / h, *cam))


def outputs(r):
    return [r.read(N.BUFFER_ACCUM).copy(), r.getTexture().copy(), r.sample_count()]


def same(a, b, what):
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        if isinstance(x, np.ndarray):
            assert_same_bits(y, x, "%s, output %d" % (what, k))
        else:
            assert x == y, (what, k, x, y)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("split", [1, 3])
def test_hit_tiles_only_equals_whole_image(gpu_ctx, oracle, kind, split):
    sc = far_scene(gpu_ctx, oracle, env=env_map(16, 8) if kind == "mcs" else None)

    def run(classes):
        r = sc.renderer(kind)
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.set_option(N.OPTION_SPLIT_STREAMS, split)
        r.reset()
        outs = []
        for k in range(6):
            r.render()
            if k in (0, 1, 5):
                outs += outputs(r)
        r.reset()                                          # a second life of the same renderer
        for k in range(3):
            r.render()
        outs += outputs(r)
        r.destroy()
        return outs

    same(run(0), run(1), "%s, %d streams: tile classes on vs off" % (kind, split))
    sc.gvol.destroy()


@pytest.mark.parametrize("kind", KINDS)
def test_against_the_oracle(gpu_ctx, oracle, kind):
    sc = far_scene(gpu_ctx, oracle)
    r = sc.renderer(kind)
    o = oracle.OracleRenderer(kind, sc.osc, sc.w, sc.h)
    r.reset(); o.reset(oracle.make_frame(sc.w, sc.h, sc.m))
    for k in range(4):
        r.render()
        o.render(to_frame(oracle, sc, r._u))
    assert_same_bits(r.getTexture().view(np.uint16), o.image_f16().view(np.uint16), "%s render buffer after 4 frames" % kind)
    assert r.sample_count() == o.samples
    r.destroy(); sc.gvol.destroy()


@pytest.mark.parametrize("kind", KINDS)
def test_matrix_change_hooks_and_reset_in_between(gpu_ctx, oracle, kind):
    def run(classes):
        sc = far_scene(gpu_ctx, oracle)
        r = sc.renderer(kind)
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.reset()
        outs = []
        r.render(); r.render(); r.render()
        r.fused = False                                    # the three hooks as separate launches: whole image, same matrix
        r.render(); r.render()
        r.fused = True
        r.render()
        outs += outputs(r)
        sc.camera.transform.localTranslation = [0.3, 0.2, 1.6]     # the camera moves, nobody calls reset(): nothing may be skipped any more
        sc.camera.transform.localRotation = [0, 0, 0, 1]
        r.render(); r.render(); r.render()
        outs += outputs(r)
        r.reset()                                          # ... until the next reset
        r.fused = False
        r.render()                                         # first pass since the reset: hooks
        r.fused = True
        r.render(); r.render(); r.render()
        outs += outputs(r)
        r.destroy(); sc.gvol.destroy()
        return outs

    same(run(0), run(1), "%s" % kind)


class OddMixMCS(vpt_amd.MCSRenderer):
    """a host that does not mix with 1/n: the ray-missing pixels' accumulators are not at a fixed point"""
    def _prepare_integrate(self):
        u = super()._prepare_integrate()
        u.mix = float(np.float32(0.37))
        return u


class OddMixDepth(vpt_amd.DepthRenderer):
    def _prepare_integrate(self):
        u = super()._prepare_integrate()
        u.mix = float(np.float32(0.37))
        return u


OddMixMCS._BASE = OddMixMCS
OddMixDepth._BASE = OddMixDepth


@pytest.mark.parametrize("cls,env_alpha", [(OddMixMCS, 255), (OddMixDepth, 255), (vpt_amd.MCSRenderer, 90)])
def test_conditions_of_the_fixed_point(gpu_ctx, oracle, cls, env_alpha):
    """MCS / Depth skip only when the first pass since the reset mixed with 1 (accumulator = frame exactly); MCS also needs an
    opaque environment (its alpha channel starts from the reset's 1)"""
    env = env_map(16, 8).copy()
    env[..., 3] = env_alpha

    def run(classes):
        sc = far_scene(gpu_ctx, oracle, env=env)
        o = {'resolution': (sc.w, sc.h), 'transform': sc.transform, 'rng': GoldenRatioRng()}
        r = cls(gpu_ctx, sc.gvol, sc.camera, sc.env, o)
        r.setTransferFunction(sc.tf)
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.reset()
        for _ in range(5):
            r.render()
        outs = outputs(r)
        r.destroy(); sc.gvol.destroy()
        return outs

    same(run(0), run(1), cls.__name__)


@pytest.mark.parametrize("kind", ["eam", "mcs"])
def test_sharded_and_sequences(gpu_ctx, oracle, kind):
    sc = far_scene(gpu_ctx, oracle, w=200, h=230)

    def run(classes, shard):
        r = sc.renderer(kind, shard=shard) if shard else sc.renderer(kind)
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.set_option(N.OPTION_SPLIT_STREAMS, 2)
        r.reset()
        r.render(); r.render()
        r.play(3, use_graph=False)
        r.play(3, use_graph=True); r.play(3, use_graph=True)
        r.render()
        r.play(4, fused=True)
        r.render(); r.render()
        outs = outputs(r) + [r.global_rows()]
        r.destroy()
        return outs

    whole = run(0, None)
    same(whole[:3], run(1, None)[:3], "%s unsharded" % kind)
    for rank in range(3):
        a = run(1, (rank, 3, 8))
        g = a[3]; ok = g >= 0
        assert_same_bits(a[0][ok], whole[0][g[ok]], "%s rank %d accumulator" % (kind, rank))
        assert_same_bits(a[1][ok], whole[1][g[ok]], "%s rank %d render buffer" % (kind, rank))
    sc.gvol.destroy()


def test_eam_full_hd_on_and_off(gpu_ctx, oracle):
    """C2's size and camera: EAM 256^3 @ 1920x1080, default camera"""
    from vpt_amd.scene import default_camera
    sc = Scene(gpu_ctx, oracle, 256, 1920, 1080, camera=default_camera(1920 / 1080), noise=48.0)

    def run(classes):
        r = sc.renderer('eam')
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.reset()
        for _ in range(8):
            r.render()
        outs = outputs(r)
        r.destroy()
        return outs

    same(run(0), run(1), "EAM 1080p")
    sc.gvol.destroy()


@pytest.mark.parametrize("kind", ["mip", "eam", "mcs", "iso", "depth"])
@pytest.mark.parametrize("size", [(256, 256), (512, 512), (816, 624)])
def test_default_stream_count_follows_the_launch_size(gpu_ctx, oracle, kind, size):
    """the library's default stream count is that of a 1080p frame and follows the launch size (vpt_internal.h split_for): a frame of a few
    hundred tiles runs its whole-image passes on more streams than its HIT-tile lists (512^2: EAM 3 -> 2, MCS 2 -> 1), a smaller one stays on
    one — every frame, with resets in between, equals the one-stream whole-image renderer's"""
    from vpt_amd.scene import default_camera
    w, h = size
    sc = Scene(gpu_ctx, oracle, 48, w, h, tf=colour_tf(32, 1), camera=default_camera(w / h), noise=40.0)

    def run(default):
        r = sc.renderer(kind)
        if not default:
            r.set_option(N.OPTION_SPLIT_STREAMS, 1); r.set_option(N.OPTION_TILE_CLASSES, 0)
        frames = []
        for rounds in range(2):
            r.reset()
            for _ in range(5):
                r.render()
                frames.append(outputs(r))
        r.play(3, use_graph=False); frames.append(outputs(r))
        r.play(4, fused=True); frames.append(outputs(r))
        r.destroy()
        return frames

    want, got = run(False), run(True)
    for k, (a, b) in enumerate(zip(want, got)):
        same(a, b, "%s %dx%d, frame %d" % (kind, w, h, k))
    sc.gvol.destroy()


# ---- render destinations other than the renderer's own buffer (round 4) ---------------------------------------------------------------
# A pass that launches the HIT tiles only leaves the other texels of its DESTINATION as they are: right only where a whole-image pass has
# written that destination since the reset.  A caller's render target, the slots of a bucket and the gather's ring change from frame to
# frame — each must take one whole pass first (TileClasses.complete in vpt_internal.h).

def reference_frames(sc, kind, n, **props):
    ref = sc.renderer(kind)
    ref.set_option(N.OPTION_TILE_CLASSES, 0)
    for k, v in props.items():
        setattr(ref, k, v)
    ref.reset()
    frames = []
    for _ in range(n):
        ref.render()
        frames.append(ref.getTexture().copy())
    ref.destroy()
    return frames


@pytest.mark.parametrize("kind", ["eam", "mip", "mcs", "iso"])
def test_render_targets_cycling_through_three_buffers(gpu_ctx, oracle, kind):
    """vpt_renderer_set_render_target into three caller-owned buffers in turn (here: the render buffers of three idle renderers), then
    the same buffers again WITHOUT announcing them anew is what play_into does — both against an unsharded renderer with classes off"""
    sc = far_scene(gpu_ctx, oracle)
    want = reference_frames(sc, kind, 9)
    holders = [sc.renderer(kind) for _ in range(3)]
    targets = [h.render_buffer_device() for h in holders]
    r = sc.renderer(kind)
    r.reset()
    for k in range(9):
        ptr, nbytes = targets[k % 3]
        r.set_render_target(ptr, nbytes)
        r.render()
        r.join()
        gpu_ctx.synchronize()
        assert_same_bits(holders[k % 3].getTexture(), want[k], "%s frame %d through a cycling render target" % (kind, k))
    r.set_render_target(0, 0)
    r.destroy()
    for h in holders:
        h.destroy()
    sc.gvol.destroy()


@pytest.mark.parametrize("kind", ["eam", "mip", "depth"])
def test_play_into_three_slots(gpu_ctx, oracle, kind):
    """vpt_renderer_play_into: frame i of a call into slot i of a bucket, the same bucket call after call (the torch.distributed
    pipeline's shape).  The bucket is the frame ring of an idle MCM renderer, read back slot by slot"""
    sc = far_scene(gpu_ctx, oracle)
    want = reference_frames(sc, kind, 9)
    holder = sc.renderer('mcm')
    holder.reset(); holder.play(3, frames=True)                   # allocates the ring; three readable slots
    import ctypes as C
    p, n = C.c_void_p(), C.c_size_t()
    N.check(N.lib().vpt_renderer_frame_ring_device(holder._h, C.byref(p), C.byref(n)))
    r = sc.renderer(kind)
    r.reset()
    for call in range(3):
        r.play_into(3, p.value, n.value)
        r.join()
        gpu_ctx.synchronize()
        for i in range(3):
            assert_same_bits(holder.read_frame_slot(i), want[3 * call + i], "%s call %d slot %d" % (kind, call, i))
    r.destroy(); holder.destroy(); sc.gvol.destroy()


@pytest.mark.parametrize("kind", ["eam", "mip"])
@pytest.mark.parametrize("root", [-1, 0])
def test_native_gather_ring_with_a_marcher(gpu_ctx, oracle, kind, root):
    """vpt_gather_render renders into send[frame % 16] (or, as the root, into its slot of recv[..]): 20 frames, so the ring wraps"""
    from vpt_amd.tiles import RcclFrameGather
    sc = far_scene(gpu_ctx, oracle)
    want = reference_frames(sc, kind, 20)
    shard = sc.renderer(kind, shard=(0, 1, 8))
    shard.reset()
    g = RcclFrameGather(shard, RcclFrameGather.unique_id(), 0, 1, root=root)
    for k in range(20):
        g.render()
        if k in (0, 1, 2, 15, 16, 17, 19):
            assert_same_bits(g.frame(), want[k], "%s gathered frame %d (root %d)" % (kind, k, root))
    g.destroy(); shard.destroy(); sc.gvol.destroy()
