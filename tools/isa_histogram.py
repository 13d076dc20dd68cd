#!/usr/bin/env python3
"""Instruction histogram of a kernel in vpt_amd/csrc/vpt_mcm{_hit,,_seq}.s (make -C vpt_amd/csrc asm).

  python tools/isa_histogram.py [mangled-name-substring] [--json out.json]

Splits the kernel into basic blocks, marks the blocks of the innermost loop that holds the most VALU instructions (the
MCM event loop), classifies every VALU instruction by issue-cost class (tools/valu_rates.hip, MI355X, >= 4 waves per SIMD):

  full   v_fma / v_fmac / v_fmaak / v_fmamk / v_mul_f32 / v_add_f32 / v_sub_f32, integer add / sub, shifts, and / or / xor,
         v_mov, v_cndmask                                                             2 cycles per wave64 instruction
  half   v_mul_lo_u32, v_mul_u32_u24, conversions, min / max / med3 / min3 / max3, fract, rndne, three-operand integer
         (add3, lshl_add, lshl_or, and_or, bfe, perm), compares, v_pk_{fma,mul,add}_f32 (two fp32 operations)   4
  trans  v_rcp / v_rsq / v_sqrt / v_log / v_exp / v_sin / v_cos                       8   (v_mad_u64_u32: 6)
(r02 measurement at 7-8 waves per SIMD: 2.37 / 4.2 / 8.25 "cycles at 2.4 GHz", i.e. 2 / 4 / 8 cycles at the ~2.0 GHz the chip
holds under this load; one dependent chain per wave reaches the same rates from 2 waves per SIMD on)

and attributes the event loop's instructions to phases by signature: a PCG round = the 9 instructions around each
`>> 28` shift; the rest is reported per basic block (the blocks of the deposit + resetPhoton path, the scattering path and the
common path are named by what they contain).  Static counts: one trip through every block of the loop; the dynamic count per
wave-event depends on which branches the wave's lanes take (PMC SQ_INSTS_VALU gives that)."""
import json
import re
import sys

COST = {"full": 2.0, "half": 4.0, "trans": 8.0}


def classify(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if re.match(r"v_(rcp|rsq|sqrt|log|exp|sin|cos)_f32", base):
        return "trans"
    if re.match(r"v_(fma|fmac|fmaak|fmamk|mul|add|sub|subrev|mac|mad)_f32$", base):
        return "full"
    if re.match(r"v_(add|sub|subrev)_(u32|i32|co_u32)$", base) or re.match(r"v_(lshrrev|lshlrev|ashrrev)_b32$", base) or \
            re.match(r"v_(and|or|xor|not)_b32$", base) or re.match(r"v_(mov_b32|mov_b64|cndmask_b32|accvgpr)", base):
        return "full"
    return "half"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    want = args[0] if args else "k_mcm_integrateILb1ELi0EE"
    out_json = None
    if "--json" in sys.argv:
        out_json = sys.argv[sys.argv.index("--json") + 1]
    lines = []
    for unit in ("vpt_mcm_hit", "vpt_mcm", "vpt_mcm_seq"):
        lines += open("vpt_amd/csrc/%s.s" % unit).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$]*:", l) and want in l.split(":")[0])
    name = lines[start].split(":")[0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], {"label": "entry", "loop": None, "ins": []}
    for l in lines[start + 1:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            blocks.append(cur)
            lm = re.search(r"in Loop: Header=(BB\d+_\d+)", l) or (re.search(r"Loop Header", l) and re.match(r"\.L(BB\d+_\d+)", l))
            loop = lm.group(1) if lm else None
            cur = {"label": m.group(1), "loop": loop, "ins": []}
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur["ins"].append(t.split()[0])
    blocks.append(cur)
    # innermost loop with the most VALU instructions
    per_loop = {}
    for b in blocks:
        if b["loop"]:
            per_loop[b["loop"]] = per_loop.get(b["loop"], 0) + sum(1 for i in b["ins"] if i.startswith("v_"))
    hot = max(per_loop, key=per_loop.get)

    def hist(bl):
        h = {"full": 0, "half": 0, "trans": 0}
        other = {"salu": 0, "vmem": 0, "lds": 0}
        ops = {}
        for b in bl:
            for i in b["ins"]:
                if i.startswith("v_"):
                    h[classify(i)] += 1
                    ops[i] = ops.get(i, 0) + 1
                elif i.startswith("s_"):
                    other["salu"] += 1
                elif i.startswith("global_") or i.startswith("buffer_"):
                    other["vmem"] += 1
                elif i.startswith("ds_"):
                    other["lds"] += 1
        return h, other, ops

    loop_blocks = [b for b in blocks if b["loop"] == hot]
    res = {"kernel": name, "source": "vpt_amd/csrc/vpt_mcm{_hit,,_seq}.s", "cost_model_cycles_per_wave_instruction": COST, "event_loop_header": hot}
    for key, bl in (("whole_kernel_static", blocks), ("event_loop_static", loop_blocks)):
        h, other, ops = hist(bl)
        res[key] = {"valu_by_class": h, "valu_total": sum(h.values()), "other": other,
                    "valu_cycles_by_model": round(sum(h[c] * COST[c] for c in h), 1),
                    "top_ops": dict(sorted(ops.items(), key=lambda kv: -kv[1])[:24])}
    pcg_rounds = 0
    per_block = []
    for b in loop_blocks:
        h, other, ops = hist([b])
        n = sum(h.values())
        if not n:
            continue
        r = sum(1 for i, l in enumerate(b["ins"]) if l == "v_lshrrev_b32")      # refined below
        per_block.append({"block": b["label"], "valu": n, "by_class": h, "cycles_by_model": round(sum(h[c] * COST[c] for c in h), 1),
                          "mul_lo_u32": ops.get("v_mul_lo_u32", 0), "trans": h["trans"], "vmem": other["vmem"], "lds": other["lds"]})
        pcg_rounds += ops.get("v_mul_lo_u32", 0) / 2.0
    res["event_loop_blocks"] = per_block
    res["pcg_rounds_in_loop_static"] = pcg_rounds
    res["pcg_valu_instructions_static"] = pcg_rounds * 9
    s = json.dumps(res, indent=1)
    if out_json:
        open(out_json, "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    main()
