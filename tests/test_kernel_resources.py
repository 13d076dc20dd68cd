"""CPU: compiles the HIP source to assembly for gfx950 (hipcc cross-compiles without a GPU) and checks the register / scratch
budget of the headline kernels from the compiler's own resource report.  Guards against a class of silent 5x regressions: an
lvalue conditional on kernel-argument members (`c ? a.ny : a.nx`) makes the whole argument block live in scratch memory —
results stay bit-exact, only the speed collapses (r02: 137 us -> 672 us per frame)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_mcm_kernels_use_no_scratch_and_fit_seven_waves():
    # (the MCM kernels live in three translation units: vpt_mcm{,_hit,_seq}.hip -> *.s + the compiler's resource remarks in *.resources.txt)
    csrc = os.path.join(ROOT, "vpt_amd", "csrc")
    units = ["vpt_mcm", "vpt_mcm_hit", "vpt_mcm_seq"]
    res = subprocess.run(["make", "-C", csrc, "-B"] + [u + ".s" for u in units], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert res.returncode == 0, res.stdout.decode()[-2000:]
    text = "".join(open(os.path.join(csrc, u + ".resources.txt")).read() for u in units)
    usage = {}
    cur = None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1); usage[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            usage[cur][m.group(1).strip()] = int(m.group(2))
    # LINEAR filter, one channel: bit-exact and fast-math, 32-bit and brick-code (> 4 GiB) tables, hooks and fused render
    hot = {k: v for k, v in usage.items() if re.match(r"_Z15k_mcm_integrateILb[01]ELi(0|1|16|17)EE", k) or k.startswith("_Z11k_mcm_multiILi0E") or k.startswith("_Z11k_mcm_multiILi16E")}
    assert len(hot) == 10, sorted(hot)
    for name, u in hot.items():
        multi = "k_mcm_multi" in name                         # the fused-pass kernels may spill a few registers around their pass loop
        assert u.get("ScratchSize", 0) <= (64 if multi else 0), (name, u)
        assert u.get("VGPRs Spill", 0) <= (16 if multi else 0), (name, u)
        assert u.get("VGPRs", 999) <= 72, (name, u)          # 7 waves per SIMD
        assert u.get("Occupancy", 0) >= 7, (name, u)
    # the frame-sequence kernels (VPT_PLAY_FRAMES) are compiled for 4 waves per SIMD and must not spill there
    frames = {k: v for k, v in usage.items() if k.startswith("_Z12k_mcm_framesILi0E") or k.startswith("_Z12k_mcm_framesILi16E")}
    assert len(frames) == 2, sorted(frames)
    for name, u in frames.items():
        assert u.get("ScratchSize", 0) == 0 and u.get("VGPRs", 999) <= 128 and u.get("Occupancy", 0) >= 4, (name, u)
    # the MISS-tile kernels of the tile classes: 8 waves per SIMD, no scratch
    miss = {k: v for k, v in usage.items() if re.match(r"_Z10k_mcm_missILb[01]ELi(0|16)ELb", k)}
    assert len(miss) == 12, sorted(miss)
    for name, u in miss.items():
        assert u.get("ScratchSize", 0) == 0 and u.get("VGPRs", 999) <= 64 and u.get("Occupancy", 0) >= 8, (name, u)
    # ... of the other volume formats (NEAREST / two channels / float texels, round 4): 14 variants x fused or not, no scratch, >= 7 waves
    other = {k: v for k, v in usage.items() if k.startswith("_Z10k_mcm_missILb") and k not in miss}
    assert len(other) == 28, sorted(other)
    for name, u in other.items():
        assert u.get("ScratchSize", 0) == 0 and u.get("Occupancy", 0) >= 7, (name, u)
    # the HIT-tile kernel sampling the column records (VPT_V_REC = 64): the same budget as the brick form
    rec = {k: v for k, v in usage.items() if re.match(r"_Z15k_mcm_integrateILb[01]ELi(64|65|80|81)EE", k)}
    assert len(rec) == 8, sorted(rec)
    for name, u in rec.items():
        assert u.get("ScratchSize", 0) == 0 and u.get("VGPRs", 999) <= 72 and u.get("Occupancy", 0) >= 7, (name, u)
