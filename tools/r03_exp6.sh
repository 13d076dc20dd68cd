mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp6.txt; : > $O
for l in b3 b4; do for sp in 2; do python3 tools/ab_mcm.py --lib gpurun_ab/$l.so --tag $l --split $sp >> $O 2>&1; done; done
for l in b3 b4; do for sp in 2; do python3 tools/ab_mcm.py --lib gpurun_ab/$l.so --tag $l --split $sp >> $O 2>&1; done; done
for sh in 3,8,8 1,4,8 0,2,8; do for sp in 1 2 3; do for cl in 0 1; do
python3 tools/ab_mcm.py --lib gpurun_ab/b4.so --tag b4 --split $sp --classes $cl --shard $sh --frames 1000 >> $O 2>&1; done; done; done
python3 tools/ab_mcm.py --lib gpurun_ab/b4.so --tag b4 --split 2 --fast 0 --shard 3,8,8 --frames 1000 >> $O 2>&1
grep -v amdgpu.ids $O
