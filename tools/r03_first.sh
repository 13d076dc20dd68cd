#!/bin/bash
# round 3, first GPU contact of the tile classes: new tests, smoke, A/B of the bench line (classes on / off, both variants)
set -o pipefail
OUT=gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_tile_classes.py -x -q -m gpu > $OUT/first_tests.log 2>&1 || { tail -40 $OUT/first_tests.log; exit 1; }
tail -3 $OUT/first_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for fm in 1 0; do for tc in 1 0; do for sp in 3 1; do
  python bench.py --fast-math $fm --tile-classes $tc --split-streams $sp --other-configs 0 --cpu-baseline 0 --stream-probe 0 > $OUT/ab_fm${fm}_tc${tc}_sp${sp}.json 2> $OUT/ab_fm${fm}_tc${tc}_sp${sp}.err || { tail -5 $OUT/ab_fm${fm}_tc${tc}_sp${sp}.err; exit 1; }
  python - <<PY
import json
j=json.load(open("$OUT/ab_fm${fm}_tc${tc}_sp${sp}.json"))
print("fast %s classes %s streams %s: %.4f ms/step frac %.3f check %s %s" % ($fm,$tc,$sp,j["ms_per_step"],j["roofline"]["frac"],j["frame_check"],j["config"].get("tile_classes")))
PY
done; done; done
