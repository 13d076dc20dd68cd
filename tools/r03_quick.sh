python -m pytest tests/test_gpu_readers.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_js_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -6
