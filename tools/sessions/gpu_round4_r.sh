#!/bin/bash
# round 4, session r: the skinny 1x1 weight gradient as a memory pass — tests, training line, per-geometry rows
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_grad_gpu.py tests/test_train_full.py tests/test_train_step.py tests/test_loss_phases.py tests/test_discriminator.py tests/test_checkpoint.py -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/round4_r_tests.log 2>&1; tail -n 5 gpurun_out/round4_r_tests.log | cut -c1-400
grep -E "^E  " gpurun_out/round4_r_tests.log | head -8 | cut -c1-1500
timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/round4_r_train.json 2>> gpurun_out/round4_r_bench.err
python -c "import json; d=json.load(open('gpurun_out/round4_r_train.json')); print('TRAIN', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/round4_r_bench.err
timeout 600 python tests/gpu_time_train_convs.py > gpurun_out/round4_r_convs.log 2>&1; grep -E "wgrad .* 1 s1 t[01]  512x512" gpurun_out/round4_r_convs.log | head -8 | cut -c1-160
