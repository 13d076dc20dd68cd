mkdir -p gpurun_out/r03/profiles
python3 tools/shard8_probe.py gpurun_out/r03/profiles/r03_shard8.json > gpurun_out/r03/shard8.log 2>&1 || { tail -5 gpurun_out/r03/shard8.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r03/shard8.log | tail -70
