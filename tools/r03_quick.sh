python -m pytest tests/test_gpu_tonemap.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -3
python3 tools/tonemap_fuse_rate.py artistic 2>&1 | grep -v amdgpu
