#!/bin/bash
# end-of-round verification on the GPU box: the whole -m gpu suite in one process, smoke(), the default bench line
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_verify; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > "$out/gpu_tests.txt" 2>&1; rc=$?; echo "exit $rc"; tail -4 "$out/gpu_tests.txt"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "test run was killed: stopping"; exit 1; fi
echo "== smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.txt" 2>&1; rc=$?; echo "exit $rc"; tail -2 "$out/smoke.txt"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "smoke was killed: stopping"; exit 1; fi
echo "== default bench line"; timeout -k 10 800 python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "exit $?"; tail -c 600 "$out/bench_default.json"; tail -2 "$out/bench_default.err"
cat gpurun_out/r02_configs.json 2>/dev/null | head -40
