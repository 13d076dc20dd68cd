mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp5.txt; : > $O
for hl in 0 53000 40000 32000 26000 22800; do for ml in 0 20000 26000; do
  VPT_EXP_HIT_LDS=$hl VPT_EXP_MISS_LDS=$ml python3 tools/ab_mcm.py --lib gpurun_ab/b3.so --tag "hitlds=$hl misslds=$ml" --split 2 --blocks 3 >> $O 2>&1
done; done
grep -v amdgpu.ids $O
