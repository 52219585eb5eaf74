#!/bin/bash
# round-6 session zx: the split-K plan's two knobs (work-groups per CU aimed at, fewest K steps per split) once more, now that the finish launch reads its partial tiles eight at a time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zx
for rep in 1 2 3; do
  for knobs in "2 4" "3 4" "4 4" "2 2" "4 2" "3 3"; do
    set -- $knobs
    P3D_SPLITK_PER_CU=$1 P3D_SPLITK_MIN_STEPS=$2 timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs --no-exact-fp32 2>/dev/null | python -c "
import json,sys; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per_cu=$1 min_steps=$2 rep $rep:', d['value'], 'img/s,', d['ms_per_step'], 'ms,', d.get('stage_ms'))"
  done
done
echo finished
