"""GPU: column records (VPT_OPTION_COLUMN_RECORDS, vpt_device.h record_addr) — the MCM renderer's in-cube samples come from a third
layout of the volume (one dword per voxel = the 2 x 2 x-y footprint of its cell, a voxel column contiguous: eight taps by ONE
dword-aligned 8-byte gather).  Same taps, same lerps: everything a caller can read must be identical, bit for bit, with the option on
and off and identical to the CPU oracle — odd sizes (clamped x + 1 / y + 1 taps, the last record of the last column), non-cubic
volumes (2-D Z-order of the columns with unequal bit counts), the wide (> 4 GiB) addressing forced onto a small volume, the boundary
atlas off (out-of-cube samples from the records too), blocks uploaded after the first pass (the records are rebuilt), tile classes
on two streams, the bucket kernels and frame sequences in one launch."""
import numpy as np
import pytest

import vpt_amd
from vpt_amd import _native as N
from vpt_amd.synthetic import colour_tf, GoldenRatioRng

from conftest import orbit_camera
from test_gpu_parity import Scene, to_frame, assert_same_bits, MCM_BUFFERS

pytestmark = pytest.mark.gpu


def buffers(r):
    return [r.read(b).copy() for b in MCM_BUFFERS] + [r.getTexture().copy()]


@pytest.mark.parametrize("dims", [(24, 24, 24), (17, 23, 9), (1, 5, 31), (40, 3, 2)])
@pytest.mark.parametrize("fast", [0, 1])
def test_records_on_and_off_give_identical_buffers(gpu_ctx, oracle, dims, fast):
    sc = Scene(gpu_ctx, oracle, 0, 176, 112, tf=colour_tf(48, 1), camera=orbit_camera(176 / 112, 0.7, -0.3, 1.9), dims=dims)

    def run(records, wide, atlas, split):
        sc.gvol.set_wide_tables(wide)
        r = sc.renderer('mcm')
        r.set_option(N.OPTION_COLUMN_RECORDS, records)
        r.set_option(N.OPTION_BOUNDARY_ATLAS, atlas)
        r.set_option(N.OPTION_FAST_MATH, fast)
        r.set_option(N.OPTION_SPLIT_STREAMS, split)
        r.extinction = 6; r.steps = 5; r.anisotropy = -0.2
        r.reset()
        for _ in range(4):
            r.render()
        outs = buffers(r)
        r.play(5, fused=True)
        outs += buffers(r)
        r.play(3, frames=True)
        outs += buffers(r)
        assert r.sample_count() == sc.w * sc.h * 5 * 12
        r.destroy()
        return outs

    want = run(0, 0, 1, 1)
    for records, wide, atlas, split in ((1, 0, 1, 1), (1, 1, 1, 2), (1, 0, 0, 2), (1, 1, 0, 1)):
        got = run(records, wide, atlas, split)
        for k, (x, y) in enumerate(zip(want, got)):
            assert_same_bits(y, x, "records %d wide %d atlas %d split %d, output %d" % (records, wide, atlas, split, k))
    sc.gvol.set_wide_tables(0)
    sc.gvol.destroy()


def test_records_against_the_oracle_and_rebuilt_after_an_upload(gpu_ctx, oracle):
    """contract arithmetic, records on (the default): every state buffer equals the oracle's; then a block is uploaded into the
    volume — the records must follow"""
    sc = Scene(gpu_ctx, oracle, 0, 160, 96, tf=colour_tf(64, 1), camera=orbit_camera(160 / 96, -0.4, 0.5, 1.8), dims=(21, 30, 27))
    r = sc.renderer('mcm')
    r.extinction = 8; r.steps = 6
    o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
    r.reset()
    o.reset(oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(GoldenRatioRng()())))
    for _ in range(3):
        r.render()
        o.render(to_frame(oracle, sc, r._u))
    for b, s in zip(MCM_BUFFERS, o.state):
        assert_same_bits(r.read(b), s.reshape(sc.h, sc.w, 4), "state buffer %d" % b)
    # a new block in the middle of the volume (texSubImage3D semantics): bricks, atlas AND records are rebuilt
    vol2 = sc.vol.copy()
    blk = (255 - vol2[5:15, 8:20, 3:17]).copy()
    vol2[5:15, 8:20, 3:17] = blk
    sc.gvol.upload_block(3, 8, 5, blk)
    osc2 = oracle.OracleScene(vol2, 'linear', tf=sc.tf)
    o2 = oracle.OracleRenderer('mcm', osc2, sc.w, sc.h)
    for k in range(4):
        o2.state[k][...] = o.state[k]
    for _ in range(3):
        r.render()
        o2.render(to_frame(oracle, sc, r._u))
    for b, s in zip(MCM_BUFFERS, o2.state):
        assert_same_bits(r.read(b), s.reshape(sc.h, sc.w, 4), "state buffer %d after the upload" % b)
    r.destroy(); sc.gvol.destroy()


def test_records_are_an_mcm_option_and_leave_other_volumes_alone(gpu_ctx, oracle):
    sc = Scene(gpu_ctx, oracle, 16, 64, 48, filt="nearest")
    r = sc.renderer('mcm')
    r.set_option(N.OPTION_COLUMN_RECORDS, 1)              # NEAREST filter: the bricks are sampled, whatever the option says
    r.reset(); r.render()
    o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
    o.reset(oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(GoldenRatioRng()())))
    o.render(to_frame(oracle, sc, r._u))
    assert_same_bits(r.read(N.BUFFER_MCM_RADIANCE), o.state[3].reshape(sc.h, sc.w, 4), "NEAREST volume")
    r.destroy()
    e = sc.renderer('eam')
    with pytest.raises(vpt_amd.VptError):
        e.set_option(N.OPTION_COLUMN_RECORDS, 1)
    e.destroy(); sc.gvol.destroy()
