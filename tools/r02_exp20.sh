#!/bin/bash
# (1) library-then-torch vs torch-then-library in one process: which order aborts at exit; (2) a 1/8 shard on 1..3 streams;
# (3) kernel times inside vpt_volume_finalize
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp20; mkdir -p "$out"
export TMPDIR=/tmp
cat > /tmp/order_a.py <<'PY'
import sys; sys.path.insert(0, '.')
import vpt_amd
ctx = vpt_amd.Context(0); ctx.synchronize()
import torch
x = torch.zeros(1024, device='cuda'); torch.cuda.synchronize()
print("library first, torch second: ran", float(x.sum()))
PY
cat > /tmp/order_b.py <<'PY'
import sys; sys.path.insert(0, '.')
import torch
x = torch.zeros(1024, device='cuda'); torch.cuda.synchronize()
import vpt_amd
ctx = vpt_amd.Context(0); ctx.synchronize()
print("torch first, library second: ran", float(x.sum()))
PY
cat > /tmp/order_c.py <<'PY'
import sys; sys.path.insert(0, '.')
import vpt_amd
ctx = vpt_amd.Context(0); ctx.synchronize()
import torch
print("library first, torch imported but unused: ran")
PY
for o in a b c; do timeout -k 5 200 python3 /tmp/order_$o.py > "$out/order_$o.txt" 2>&1; echo "order $o exit $?"; tail -2 "$out/order_$o.txt"; done
echo "== new test alone"; timeout -k 5 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "caller_owned or split" > "$out/t1.txt" 2>&1; echo "exit $?"; tail -3 "$out/t1.txt"
echo "== shard split probe"; timeout -k 5 600 python3 tools/shard_split_probe.py > "$out/shard_split.json" 2>"$out/shard_split.err"; cat "$out/shard_split.json"
echo "== finalize kernels"; cd /tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python3 $root/tools/brickify_rate.py > "$out/kt.log" 2>&1; cd $root
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg", round(float(r["MaxNs"]) / 1e3, 1), "us max")
PY
