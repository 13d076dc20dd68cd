"""CPU: loaders and readers (vpt_amd/loaders.py, readers.py; js/vpt/loaders, js/vpt/readers) against what the reference's
own RAWReader.js / ZIPReader.js / BVPReader.js / ReaderFactory.js returned for the same bytes
(tests/golden/readers_r01.json, produced by running them under node: tests/golden/make_reader_fixture.py)."""
import base64
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import vpt_amd
from vpt_amd.loaders import BlobLoader, FileLoader, LoaderFactory
from vpt_amd.readers import RAWReader, ZIPReader, BVPReader, ReaderFactory

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def fx():
    d = json.load(open(os.path.join(HERE, "golden", "readers_r01.json")))
    d["archive"] = base64.b64decode(d["archive_base64"]); d["raw"] = base64.b64decode(d["raw_base64"])
    return d


def digest(data):
    b = bytes(data)
    return {"length": len(b), "sha256": hashlib.sha256(b).hexdigest()}


@pytest.mark.parametrize("loader_kind", ["blob", "file"])
def test_zip_reader_matches_reference(fx, tmp_path, loader_kind):
    ref = fx["reference"]
    if loader_kind == "file":
        p = tmp_path / "a.bvp"; p.write_bytes(fx["archive"])
        loader = FileLoader(str(p))
    else:
        loader = BlobLoader(fx["archive"])
    z = ZIPReader(loader)
    assert z.getFiles() == ref["zip_files"]
    assert z._cd == ref["zip_cd"]                                   # every parsed central-directory field (ZIPReader.js:71-90)
    for name in ref["zip_files"]:
        assert digest(z.readFile(name)) == ref["zip_file_digests"][name]
    with pytest.raises(RuntimeError) as e:
        z.readFile('missing.bin')
    assert str(e.value) == ref["zip_missing"]                       # ZIPReader.js:28


def test_bvp_reader_matches_reference(fx):
    ref = fx["reference"]
    r = BVPReader(BlobLoader(fx["archive"]))
    assert r.readMetadata() == ref["bvp_metadata"]
    for i, want in enumerate(ref["bvp_blocks"]):
        assert digest(r.readBlock(i)) == want
    r2 = BVPReader(BlobLoader(fx["archive"]))                        # readBlock before readMetadata (BVPReader.js:23-25)
    assert digest(r2.readBlock(1)) == ref["bvp_blocks"][1]


def test_raw_reader_matches_reference(fx):
    ref = fx["reference"]
    w, h, d = fx["raw_dims_whd"]
    r = RAWReader(BlobLoader(fx["raw"]), {'width': w, 'height': h, 'depth': d})
    assert r.readMetadata() == ref["raw_metadata"]
    for i, want in enumerate(ref["raw_blocks"]):
        assert digest(r.readBlock(i)) == want
    assert RAWReader(BlobLoader(b"")).readMetadata()["modalities"][0]["dimensions"] == {"width": 0, "height": 0, "depth": 0}   # RAWReader.js:8-12


def test_factories(fx):
    assert fx["reference"]["factory"] == {"bvp": True, "raw": True, "zip": True}
    assert ReaderFactory('bvp') is BVPReader and ReaderFactory('raw') is RAWReader and ReaderFactory('zip') is ZIPReader
    with pytest.raises(RuntimeError) as e:
        ReaderFactory('nrrd')
    assert str(e.value) == fx["reference"]["factory_unknown"]
    assert LoaderFactory('blob') is BlobLoader
    with pytest.raises(RuntimeError, match='No suitable class'):
        LoaderFactory('ftp')


def test_block_placement_reassembles_the_volume(fx):
    """Volume.readModality's texSubImage3D placements (Volume.js:63-71) put the BVP blocks back where they were cut"""
    w, h, d = fx["archive_dims_whd"]
    want = np.array(fx["archive_volume_zyx"], dtype=np.uint8).reshape(d, h, w)
    r = BVPReader(BlobLoader(fx["archive"]))
    md = r.readMetadata()
    mod = md["modalities"][0]
    got = np.zeros((d, h, w), dtype=np.uint8)
    for pl in mod["placements"]:
        bd = md["blocks"][pl["index"]]["dimensions"]
        blk = np.frombuffer(bytes(r.readBlock(pl["index"])), dtype=np.uint8).reshape(bd["depth"], bd["height"], bd["width"])
        x, y, z = pl["position"]["x"], pl["position"]["y"], pl["position"]["z"]
        got[z:z + bd["depth"], y:y + bd["height"], x:x + bd["width"]] = blk
    assert (got == want).all()


def test_file_loader_ranges(tmp_path):
    data = np.arange(1000, dtype=np.uint32).tobytes()
    p = tmp_path / "blob.bin"; p.write_bytes(data)
    f, b = FileLoader(str(p)), BlobLoader(data)
    assert f.readLength() == b.readLength() == len(data)
    for s, e in ((0, 10), (5, 5), (3990, 4000), (3990, 5000), (4000, 4010), (17, 2049)):
        assert bytes(f.readData(s, e)) == bytes(b.readData(s, e)) == data[s:e]
    f.close()


def test_js_readers_match_reference():
    """the Node host's readers over the same fixture"""
    out = subprocess.run(["node", os.path.join(ROOT, "js", "test", "test_readers.js")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0, out.stderr.decode() + out.stdout.decode()
    assert b"js readers ok" in out.stdout
