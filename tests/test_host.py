"""CPU: host-side mirror of the reference's plugin surface (no device calls)."""
import numpy as np
import pytest

import vpt_amd
from vpt_amd.property_bag import PropertyBag, CustomEvent
from vpt_amd.scene import Node, Transform, PerspectiveCamera, mat4, default_camera
from vpt_amd.volume import RAWReader, GL_RED, GL_R8, GL_UNSIGNED_BYTE
from vpt_amd import tiles


def test_property_bag_registers_attributes_and_dispatches():
    bag = PropertyBag()
    bag.registerProperties([{'name': 'steps', 'value': 64}, {'name': 'extinction', 'value': 1}])
    assert bag.steps == 64 and bag.extinction == 1 and len(bag.properties) == 2
    seen = []
    bag.addEventListener('change', lambda e: seen.append(e.detail))
    bag.dispatchEvent(CustomEvent('change', {'detail': {'name': 'steps', 'value': 8}}))
    assert seen == [{'name': 'steps', 'value': 8}]


def test_renderer_factory_names():
    assert vpt_amd.RendererFactory('mip') is vpt_amd.MIPRenderer
    assert vpt_amd.RendererFactory('eam') is vpt_amd.EAMRenderer
    assert vpt_amd.RendererFactory('mcs') is vpt_amd.MCSRenderer
    assert vpt_amd.RendererFactory('mcm') is vpt_amd.MCMRenderer
    assert vpt_amd.RendererFactory('iso') is vpt_amd.ISORenderer and vpt_amd.RendererFactory('depth') is vpt_amd.DepthRenderer
    assert vpt_amd.RendererFactory('lao') is vpt_amd.LAORenderer and vpt_amd.RendererFactory('dos') is vpt_amd.DOSRenderer
    for name in ('nope', ''):
        with pytest.raises(RuntimeError, match='No suitable class'):      # RendererFactory.js:21
            vpt_amd.RendererFactory(name)


def test_transform_events_and_matrices():
    t = Transform(Node())
    hits = []
    t.addEventListener('change', lambda e: hits.append(e.type))
    t.localTranslation = [1, 2, 3]
    t.localScale = [2, 2, 2]
    assert hits == ['change', 'change'] and t.version == 2
    m = t.localMatrix
    assert m[12] == 1 and m[13] == 2 and m[14] == 3 and m[0] == 2
    inv = t.inverseLocalMatrix
    prod = mat4.multiply(mat4.create(), m, inv)
    assert np.allclose(prod, mat4.create(), atol=1e-6)


def test_default_camera_matches_rendering_context():
    cam = default_camera(16 / 9)                      # RenderingContext.js:38-40,121
    pc = cam.getComponent(PerspectiveCamera)
    assert (pc.fovy, pc.near, pc.far) == (1, 0.1, 100) and pc.aspect == 16 / 9
    assert list(cam.transform.localTranslation) == [0, 0, 2]


def test_raw_reader_metadata_shape():
    data = np.arange(4 * 3 * 5, dtype=np.uint8)
    rd = RAWReader(data, {'width': 4, 'height': 3, 'depth': 5})
    md = rd.readMetadata()                            # RAWReader.js:15-63
    mod = md['modalities'][0]
    assert mod['name'] == 'default' and mod['dimensions'] == {'width': 4, 'height': 3, 'depth': 5}
    assert (mod['format'], mod['internalFormat'], mod['type']) == (GL_RED, GL_R8, GL_UNSIGNED_BYTE)
    assert len(mod['placements']) == 5 and mod['placements'][3] == {'index': 3, 'position': {'x': 0, 'y': 0, 'z': 3}}
    assert md['blocks'][0]['dimensions'] == {'width': 4, 'height': 3, 'depth': 1}
    assert (rd.readBlock(2) == data[24:36]).all()


@pytest.mark.parametrize("height,world,rows", [(1080, 8, 8), (1080, 2, 8), (70, 3, 5), (2160, 8, 8), (17, 4, 8), (64, 1, 8)])
def test_row_sharding_is_a_partition(height, world, rows):
    rank, lrow = tiles.row_owner(height, world, rows)
    lr = tiles.local_rows(height, world, rows)
    assert lrow.max() < lr
    pairs = set(zip(rank.tolist(), lrow.tolist()))
    assert len(pairs) == height                       # every global row owned exactly once
    idx = tiles.gather_index(height, world, rows)
    assert len(set(idx.tolist())) == height and idx.max() < world * lr
    if world > 1:
        counts = np.bincount(rank, minlength=world)
        assert counts.max() - counts.min() <= rows    # balanced to within one block


def test_golden_ratio_rng_is_deterministic():
    from vpt_amd.synthetic import GoldenRatioRng
    a, b = GoldenRatioRng(), GoldenRatioRng()
    va = [a() for _ in range(5)]
    assert va == [b() for _ in range(5)] and all(0 <= v < 1 for v in va)
    assert abs(va[0] - 0.61803398875) < 1e-12


def test_buffer_spec_hooks_describe_the_reference_attachments():
    """AbstractRenderer.js:134-155 + the subclasses' _get*BufferSpec: one dict per attachment, the reference's GL enums (checked
    here against the enum values and attachment counts read from the reference's renderer classes)"""
    from vpt_amd import renderers as R
    from vpt_amd import _native as N
    GL = R.AbstractRenderer._GL
    want = {N.RENDERER_MIP: (1, 1, GL['R8']), N.RENDERER_EAM: (1, 1, GL['RGBA']), N.RENDERER_MCS: (1, 1, GL['RGBA32F']),
            N.RENDERER_MCM: (1, 4, GL['RGBA32F']), N.RENDERER_ISO: (1, 1, GL['RGBA16F']), N.RENDERER_DEPTH: (1, 1, GL['R32F']),
            N.RENDERER_LAO: (1, 1, GL['RGBA']), N.RENDERER_DOS: (1, 2, GL['RGBA32F'])}
    for cls in (R.MIPRenderer, R.EAMRenderer, R.MCSRenderer, R.MCMRenderer, R.ISORenderer, R.DepthRenderer, R.LAORenderer, R.DOSRenderer):
        r = cls.__new__(cls)
        r._resolution = (40, 30)
        nf, na, iformat = want[cls._KIND]
        fs, acc, ren = r._getFrameBufferSpec(), r._getAccumulationBufferSpec(), r._getRenderBufferSpec()
        assert len(fs) == nf and len(acc) == na and len(ren) == 1
        assert acc[0]['iformat'] == iformat and acc[0]['width'] == 40 and acc[0]['height'] == 30 and acc[0]['min'] == GL['NEAREST']
        assert ren[0]['iformat'] == GL['RGBA16F'] and ren[0]['type'] == GL['FLOAT'] and ren[0]['wrapS'] == GL['CLAMP_TO_EDGE']
    assert R.DOSRenderer.__new__(R.DOSRenderer)._BUFFER_FORMATS[N.RENDERER_DOS][1][1] == ('RED', 'R32F', 'FLOAT')
