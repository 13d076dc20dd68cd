#!/usr/bin/env python3
"""Generates tests/golden/contract_r01.json: SHA-256 digests of the CPU oracle's buffers for small seeded scenes of the
four renderers.  The numeric contract fixes every bit of these buffers; the digests freeze it so that a
later change of either the oracle or the kernels (both must move together) is visible in review.  Inputs are
synthetic (vpt_amd.synthetic) and deterministic; nothing of the reference is read.
Run from the repository root:  python tests/golden/make_contract_fixture.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def scenes():
    from vpt_amd.synthetic import sphere_volume, colour_tf
    from conftest import default_matrix
    vol = sphere_volume(32, noise=40.0)
    tf = colour_tf(64, 1)
    return [
        dict(name="mip_linear", kind="mip", vol=vol, filter="linear", tf=None, w=96, h=64, frames=3, kw=dict(steps=40)),
        dict(name="mip_nearest", kind="mip", vol=vol, filter="nearest", tf=None, w=96, h=64, frames=2, kw=dict(steps=64)),
        dict(name="eam", kind="eam", vol=vol, filter="linear", tf=tf, w=96, h=64, frames=3, kw=dict(steps=48, extinction=60.0)),
        dict(name="mcs", kind="mcs", vol=vol, filter="linear", tf=tf, w=96, h=64, frames=3, kw=dict(extinction=12.0, light_dir=(0.48, 0.6, 0.64))),
        dict(name="mcm_iso", kind="mcm", vol=vol, filter="linear", tf=tf, w=96, h=64, frames=4, kw=dict(extinction=8.0, anisotropy=0.0, max_bounces=3, mcm_steps=6)),
        dict(name="iso", kind="iso", vol=vol, filter="linear", tf=tf, w=96, h=64, frames=3,
             kw=dict(steps=40, mcm_steps=40, isovalue=0.35, gradient_step=0.005, light_dir=(0.32444283, -0.48666424, -0.81110704))),
        dict(name="depth", kind="depth", vol=vol, filter="linear", tf=tf, w=96, h=64, frames=3, kw=dict(steps=48, extinction=60.0, threshold=0.15)),
        dict(name="lao", kind="lao", vol=vol, filter="linear", tf=tf, w=96, h=64, frames=1, kw=dict(steps=24, extinction=80.0)),
        dict(name="dos", kind="dos", vol=vol, filter="linear", tf=tf, w=96, h=64, frames=2, kw=dict(steps=50, extinction=60.0)),
        dict(name="mcm_hg", kind="mcm", vol=vol, filter="linear", tf=tf, w=96, h=64, frames=4, kw=dict(extinction=8.0, anisotropy=0.5, max_bounces=3, mcm_steps=6)),
    ], default_matrix


def dos_inputs(frame):
    """the DOS scene's explicit per-slice uniforms (uOcclusionScale.xy, uDepth) for render() call `frame`, and its six
    occlusion samples"""
    seq = lambda k: (k * 0.61803398875) % 1.0
    samples = np.array([[2 * seq(2 * i + 1) - 1, 2 * seq(2 * i + 2) - 1] for i in range(6)], np.float32)
    depths = np.linspace(0.866, 0.923, 16, dtype=np.float32)[8 * frame:8 * frame + 8]       # the cube spans NDC depth 0.8685 .. 0.9219
    slices = np.array([[0.021, 0.033, d] for d in depths], np.float32)
    return slices, samples


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run(oracle, sc, default_matrix):
    osc = oracle.OracleScene(sc["vol"], sc["filter"], tf=sc["tf"])
    m = default_matrix(sc["w"] / sc["h"])
    o = oracle.OracleRenderer(sc["kind"], osc, sc["w"], sc["h"])
    seq = lambda k: float(np.float32((k * 0.61803398875) % 1.0))
    fr = oracle.make_frame(sc["w"], sc["h"], m, seed=seq(1), **sc["kw"])
    o.reset(fr)
    for k in range(sc["frames"]):
        fr.seed = seq(k + 2); fr.offset = seq(k + 2); fr.mix = float(np.float32(1.0 / (k + 1)))
        if sc["kind"] == "dos":
            o.integrate_slices(fr, *dos_inputs(k)); o.render_frame(fr)
        else:
            o.render(fr)
    bufs = o.state if sc["kind"] == "mcm" else [o.color[o.cur], o.occlusion[o.cur]] if sc["kind"] == "dos" else [o.acc]
    return {"buffers": digest(*bufs), "render_f16": digest(o.out), "samples": int(o.samples)}


def main():
    from oracle import oracle
    sc_list, default_matrix = scenes()
    out = {"generator": "tests/golden/make_contract_fixture.py", "contract": "DESIGN.md §3 as of round 1", "scenes": {}}
    for sc in sc_list:
        out["scenes"][sc["name"]] = run(oracle, sc, default_matrix)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "contract_r01.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
