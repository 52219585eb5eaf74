#!/bin/bash
# split-K plan of the low-resolution layers (csrc/conv2d.hip splitk_plan): work-groups per CU x fewest K steps per split, benchmark step for each
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for cfg in "2 4" "1 4" "3 4" "2 8" "1 8" "2 2" "2 4"; do
  set -- $cfg
  P3D_SPLITK_PER_CU=$1 P3D_SPLITK_MIN_STEPS=$2 timeout 200 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.readline()); print('per_cu=$1 min_steps=$2', d['value'], d['ms_per_step'], d['stage_ms'])"
done 2>&1 | tee gpurun_out/splitk_sweep.log
