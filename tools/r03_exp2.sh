mkdir -p gpurun_out/r03
O=gpurun_out/r03/exp2.txt; : > $O
for round in 1 2; do
for lay in "2:" "3:" "3:2,1" "4:2,2" "4:3,1" "4:1,3" "5:3,2" "6:4,2" "6:3,3"; do
  sp=${lay%%:*}; l=${lay#*:}
  VPT_EXP_LAYOUT=$l python3 tools/ab_mcm.py --lib gpurun_ab/b2.so --tag "b2 layout=$l" --split $sp >> $O 2>&1
done
VPT_EXP_LAYOUT= python3 tools/ab_mcm.py --lib gpurun_ab/b2_sidelow.so --tag "sidelow" --split 2 >> $O 2>&1
VPT_EXP_LAYOUT=2,1 python3 tools/ab_mcm.py --lib gpurun_ab/b2_sidelow.so --tag "sidelow 2,1" --split 3 >> $O 2>&1
done
python3 tools/ab_mcm.py --lib gpurun_ab/b2.so --tag "b2" --split 1 >> $O 2>&1
python3 tools/ab_mcm.py --lib gpurun_ab/b2.so --tag "b2" --split 2 --fast 0 >> $O 2>&1
grep -v amdgpu.ids $O
