#!/bin/bash
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp35; mkdir -p "$out"
export TMPDIR=/tmp
echo "== parity"; timeout -k 5 900 python3 -m pytest tests/test_gpu_parity.py tests/test_js_gpu.py -x -q -m gpu > "$out/parity.txt" 2>&1; echo "exit $?"; tail -3 "$out/parity.txt"
