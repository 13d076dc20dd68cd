mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_fast_math.py -x -q -m gpu -s > gpurun_out/r03/t1.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r03/t1.log | tail -25; exit $rc
