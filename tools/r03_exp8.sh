mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp8.txt; : > $O
for round in 1 2; do for l in exp ballot ballot_prio; do
  python3 tools/ab_mcm.py --lib gpurun_ab/$l.so --tag $l --split 2 >> $O 2>&1
done; done
python3 tools/ab_mcm.py --lib gpurun_ab/ballot.so --tag ballot --split 2 --fast 0 >> $O 2>&1
python3 tools/ab_mcm.py --lib gpurun_ab/ballot.so --tag ballot --split 2 --shard 3,8,8 --frames 1000 >> $O 2>&1
grep -v amdgpu.ids $O
