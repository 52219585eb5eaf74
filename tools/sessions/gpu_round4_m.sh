#!/bin/bash
# round 4, session m: the weight gradient's small-channel mode — tests, the training line, the per-geometry table again
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_grad_gpu.py tests/test_ops_gpu.py tests/test_train_full.py tests/test_train_step.py tests/test_loss_phases.py tests/test_discriminator.py -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/round4_m_tests.log 2>&1; tail -n 5 gpurun_out/round4_m_tests.log | cut -c1-400
grep -E "^E  " gpurun_out/round4_m_tests.log | head -8 | cut -c1-1500
timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/round4_m_train.json 2>> gpurun_out/round4_m_bench.err
python -c "import json; d=json.load(open('gpurun_out/round4_m_train.json')); print('TRAIN', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/round4_m_bench.err
timeout 600 python tests/gpu_time_train_convs.py > gpurun_out/round4_m_convs.log 2>&1; head -n 24 gpurun_out/round4_m_convs.log | cut -c1-160
