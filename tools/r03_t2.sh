mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_marcher_classes.py tests/test_gpu_tile_classes.py -x -q -m gpu > gpurun_out/r03/t2.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r03/t2.log | tail -30
[ $rc -ne 0 ] && exit $rc
for k in eam mip iso depth mcs; do for tc in 0 1; do for sp in 1 3; do
python3 tools/ab_mcm.py --renderer $k --volume 256 --classes $tc --split $sp --frames 200 --blocks 3 --tag "$k 256" 2>&1 | grep -v amdgpu.ids
done; done; done
