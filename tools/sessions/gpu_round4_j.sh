#!/bin/bash
# round 4, session j: col_dot / skinny_contract with more loads in flight — the tests that cover them, then the training line
tag=${1:-round4_j}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bcast_gpu.py tests/test_conv_grad_gpu.py tests/test_train_full.py tests/test_train_step.py tests/test_loss_phases.py tests/test_discriminator.py tests/test_checkpoint.py -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1; tail -n 5 gpurun_out/${tag}_tests.log | cut -c1-400
grep -E "^E  " gpurun_out/${tag}_tests.log | head -8 | cut -c1-1500
timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/${tag}_train.json 2>> gpurun_out/${tag}_bench.err
python -c "import json; d=json.load(open('gpurun_out/${tag}_train.json')); print('TRAIN', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/${tag}_bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o t -- python $GRAFT_REPO_ROOT/bench.py --train-step --steps 3 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_line_train_profiled.json 2> /tmp/prof_train.err)
cp $(find /tmp/prof_train -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_train_kernel_stats.csv; head -n 14 gpurun_out/${tag}_train_kernel_stats.csv | cut -c1-150
grep -E "col_dot|skinny_contract|upsample_gen2d|fir4_cl|upfirdn2d_cl" gpurun_out/${tag}_train_kernel_stats.csv | cut -c1-200
rm -rf /tmp/prof_train
