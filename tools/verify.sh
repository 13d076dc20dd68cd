#!/bin/bash
# One GPU box: the whole -m gpu suite, smoke(), the default bench line (as the driver runs it) and the Node host's bench — what a round ends with.
#   gpurun --timeout 1200 -- 'bash tools/verify.sh r04'      -> gpurun_out/<round>/{tests_full.log,smoke.log,bench_default.json,js_bench.json}
set -o pipefail
round=${1:-r04}; out=gpurun_out/$round; mkdir -p $out
timeout -k 10 900 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu | tail -15 | tee $out/tests_full.log
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tee $out/smoke.log
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/bench_default.json'))
print('bench:', d['ms_per_step'], d['roofline']['frac'], d['frame_check'], [ (k['name'][:16], round(k['alone_us'],1), round(k['frac_alg'],3), k['traffic_ratio'] and round(k['traffic_ratio'],2)) for k in d['roofline'].get('per_kernel',[])])
o=d['other_configs']
for k in ('H_mcm_512_1080p_bit_exact','C4_mcm_1024_1080p_fast_math','C4_mcm_1024_1080p','C2_eam_256_1080p','C3_mcs_512_1080p'):
    print(k, o[k]['ms_per_frame'], (o[k].get('roofline') or o[k]).get('frac'))
"
timeout -k 10 300 node js/bench.js 2>/dev/null | tail -1 | tee $out/js_bench.json
timeout -k 10 300 node js/bench.js --fast-math 1 2>/dev/null | tail -1 | tee $out/js_bench_fast.json
