tools/r03_pmc.sh fast_classes_one_stream --fast-math 1 --split-streams 1 2>&1 | grep -v amdgpu
tools/r03_pmc.sh fast_classes_two_streams --fast-math 1 --split-streams 2 2>&1 | grep -v amdgpu
