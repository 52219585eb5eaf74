#!/bin/bash
# round 3, session c: the whole -m gpu suite on the tweaked ray-marcher + stream changes, then the benchmark with the skip-image stream and the
# two-SR-head streams on / off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/r3c_gputest.log 2>&1; tail -12 gpurun_out/r3c_gputest.log
for cfg in "1 1" "0 0" "1 0" "0 1"; do set -- $cfg
  echo "== IMG_STREAM=$1 SR_STREAMS=$2"; P3D_IMG_STREAM=$1 P3D_SR_STREAMS=$2 timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'img/s', d['ms_per_step'], 'ms', d['stage_ms'], 'render', d['roofline']['ms_per_launch'], d['config']['launch'])"
done 2>&1 | tee gpurun_out/r3c_streams.log
