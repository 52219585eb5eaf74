#!/bin/bash
# round-4 session y: Conv2dLayer's one-launch training form — its own test, the suites that pin the discriminators / loss phases to the reference, the training line with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_layer_gpu.py tests/test_conv_grad_gpu.py -m gpu -q --tb=short > gpurun_out/y_pytest1.log 2>&1; echo "pytest exit $?" >> gpurun_out/y_pytest1.log
tail -6 gpurun_out/y_pytest1.log | cut -c1-1500
timeout 900 python -m pytest tests/test_discriminator.py tests/test_loss_phases.py tests/test_train_full.py tests/test_train_step.py tests/test_dp_two_ranks_gpu.py tests/test_checkpoint.py -m gpu -q --tb=short > gpurun_out/y_pytest2.log 2>&1; echo "pytest exit $?" >> gpurun_out/y_pytest2.log
tail -4 gpurun_out/y_pytest2.log | cut -c1-1200
timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/y_train.json 2>> gpurun_out/y_bench.err
python -c "import json; d=json.load(open('gpurun_out/y_train.json')); print('TRAIN', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/y_bench.err
P3D_CONV_LAYER=0 timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/y_train_off.json 2>> gpurun_out/y_bench.err
python -c "import json; d=json.load(open('gpurun_out/y_train_off.json')); print('TRAIN unfused', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/y_bench.err
