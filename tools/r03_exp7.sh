mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp7.txt; : > $O
for hl in 0 60000 45000 34000 28000; do for ml in 0 4000; do
  VPT_EXP_HIT_LDS=$hl VPT_EXP_MISS_LDS=$ml python3 tools/ab_mcm.py --lib gpurun_ab/exp.so --tag "hitlds=$hl misslds=$ml" --split 2 --blocks 3 >> $O 2>&1
done; done
grep -v amdgpu.ids $O
