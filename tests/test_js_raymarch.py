"""CPU: the scalar JS ray-march (oracle/js/raymarch.js) — config C1 of BASELINE.json ("MIP renderer, 64^3 synthetic
sphere, 256x256, 1 frame on scalar JS CPU ray-march (plumbing, no GPU)") and a small MCM run — against the C oracle.
The JS emulates fmaf through binary64 (double rounding about once in 2^29 operations), so a stray last-bit difference
is tolerated on isolated pixels; everything else is bit-identical."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import default_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node not installed")


def run_js(tmp_path, job, vol):
    vol.tofile(tmp_path / "vol.raw")
    job = dict(job, volume=str(tmp_path / "vol.raw"), output=str(tmp_path / "out.bin"))
    (tmp_path / "job.json").write_text(json.dumps(job))
    res = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "raymarch.js"), str(tmp_path / "job.json")],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()
    return json.loads(res.stdout.decode().strip().splitlines()[-1]), (tmp_path / "out.bin").read_bytes()


def test_config_c1_mip_64_cubed_256x256_js_vs_oracle(oracle, tmp_path):
    from vpt_amd.synthetic import sphere_volume
    n, w, h, steps = 64, 256, 256, 64
    vol = sphere_volume(n)
    m = default_matrix(1.0)
    offset = float(np.float32(0.61803398875))
    job = {"kind": "mip", "nx": n, "ny": n, "nz": n, "width": w, "height": h, "steps": steps, "frames": 1, "offsets": [offset],
           "mvp_inverse_bits": m.view(np.uint32).tolist(), "filter": "linear"}
    info, raw = run_js(tmp_path, job, vol)
    got = np.frombuffer(raw, np.uint8).reshape(h, w)
    sc = oracle.OracleScene(vol, "linear")
    o = oracle.OracleRenderer("mip", sc, w, h)
    fr = oracle.make_frame(w, h, m, offset=offset, steps=steps)
    o.reset(fr); o.render(fr)
    want = o.acc.reshape(h, w)
    assert (got != want).mean() <= 1e-4
    assert info["samples"] == o.samples and got.max() == 255


def test_mcm_js_vs_oracle(oracle, tmp_path):
    from vpt_amd.synthetic import sphere_volume
    n, w, h = 24, 48, 32
    vol = sphere_volume(n, noise=40.0)
    m = default_matrix(w / h)
    seeds = [float(np.float32((k + 2) * 0.61803398875 % 1.0)) for k in range(3)]
    reset_seed = float(np.float32(0.61803398875))
    for g in (0.0, 0.5):
        job = {"kind": "mcm", "nx": n, "ny": n, "nz": n, "width": w, "height": h, "steps": 6, "bounces": 3, "extinction": 7.0,
               "anisotropy": g, "reset_seed": reset_seed, "seeds": seeds, "mvp_inverse_bits": m.view(np.uint32).tolist()}
        info, raw = run_js(tmp_path, job, vol)
        got = np.frombuffer(raw, np.float32).reshape(4, h, w, 4)
        sc = oracle.OracleScene(vol, "linear")
        o = oracle.OracleRenderer("mcm", sc, w, h)
        fr = oracle.make_frame(w, h, m, seed=reset_seed, extinction=7.0, anisotropy=g, max_bounces=3, mcm_steps=6)
        o.reset(fr)
        for s in seeds:
            fr.seed = s
            o.integrate(fr)
        want = np.stack([s.reshape(h, w, 4) for s in o.state])
        bad = (got.view(np.uint32) != want.view(np.uint32)).any(axis=(0, 3))
        assert bad.mean() <= 2e-3, bad.mean()          # an fma double-rounding flips one decision -> that pixel's path diverges
        assert info["samples"] == w * h * 6 * len(seeds)
