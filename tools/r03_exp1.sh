mkdir -p gpurun_out/r03
tools/ab.sh "base hitprio3 missprio3 sidelow miss6 miss4 eps001 hit5" 2 > gpurun_out/r03/exp1.txt 2>&1
for sp in 2 4; do python3 tools/ab_mcm.py --lib gpurun_ab/base.so --tag base --split $sp >> gpurun_out/r03/exp1.txt 2>&1; done
python3 tools/ab_mcm.py --lib gpurun_ab/base.so --tag base --fast 0 >> gpurun_out/r03/exp1.txt 2>&1
cat gpurun_out/r03/exp1.txt
