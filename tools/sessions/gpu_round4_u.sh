#!/bin/bash
# round-4 session u: the 64-channel form of the weight-gradient kernel — parity, then the rows it changes, then the training line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/wgrad_variants.txt
timeout 600 python -m pytest tests/test_conv_grad_gpu.py -m gpu -q > gpurun_out/u_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/u_pytest.log
tail -5 gpurun_out/u_pytest.log | cut -c1-600
timeout 200 python tests/gpu_time_wgrad.py small64 > /dev/null 2>gpurun_out/u_err.log
P3D_WGRAD_NO_SMALL=1 timeout 200 python tests/gpu_time_wgrad.py no_small64 > /dev/null 2>>gpurun_out/u_err.log
grep "64x64" gpurun_out/wgrad_variants.txt; grep -v amdgpu.ids gpurun_out/u_err.log | tail -5
timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/u_train.json 2>> gpurun_out/u_bench.err
python -c "import json; d=json.load(open('gpurun_out/u_train.json')); print('TRAIN', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/u_bench.err
