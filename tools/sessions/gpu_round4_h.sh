#!/bin/bash
# round 4, session h: the native up-sizing of D's raw input — tests, then the training line with and without it
tag=${1:-round4_h}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_helpers.py tests/test_discriminator.py tests/test_loss_phases.py tests/test_train_full.py -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1; tail -n 5 gpurun_out/${tag}_tests.log | cut -c1-400
grep -E "^E  " gpurun_out/${tag}_tests.log | head -8 | cut -c1-2000
for v in 1 0; do
  P3D_NATIVE_UPSIZE=$v timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/${tag}_train_upsize$v.json 2>> gpurun_out/${tag}_bench.err
  python -c "import json; d=json.load(open('gpurun_out/${tag}_train_upsize$v.json')); print('NATIVE_UPSIZE=$v', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/${tag}_bench.err
done
