mkdir -p gpurun_out/r03
tools/r03_trace.sh split1 --split-streams 1 2>&1 | grep -v amdgpu.ids | tail -12
tools/r03_trace.sh split2 --split-streams 2 2>&1 | grep -v amdgpu.ids | tail -12
