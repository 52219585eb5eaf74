#!/bin/bash
# round-4 session af: fp16 weight gradient on transposing LDS reads — parity, rows with and without it, the training line both ways
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/wgrad_variants.txt
timeout 600 python -m pytest tests/test_conv_grad_gpu.py -m gpu -q --tb=short > gpurun_out/af_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/af_pytest.log
tail -8 gpurun_out/af_pytest.log | cut -c1-800
timeout 200 python tests/gpu_time_wgrad.py tr > /dev/null 2>gpurun_out/af_err.log
P3D_WGRAD_NO_TR=1 timeout 200 python tests/gpu_time_wgrad.py no_tr > /dev/null 2>>gpurun_out/af_err.log
grep float16 gpurun_out/wgrad_variants.txt
timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/af_train.json 2>> gpurun_out/af_bench.err
python -c "import json; d=json.load(open('gpurun_out/af_train.json')); print('TRAIN', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/af_bench.err
P3D_WGRAD_NO_TR=1 timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/af_train_off.json 2>> gpurun_out/af_bench.err
python -c "import json; d=json.load(open('gpurun_out/af_train_off.json')); print('TRAIN no tr', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/af_bench.err
