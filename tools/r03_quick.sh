python -m pytest tests/test_gpu_tile_classes.py tests/test_gpu_fast_math.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -3
