#!/bin/bash
# round-4 session v: the generic kernel's 64-column form — parity on every convolution test, then the training line with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_grad_gpu.py tests/test_conv_gpu.py tests/test_split_acts.py tests/test_discriminator.py -m gpu -q > gpurun_out/v_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/v_pytest.log
tail -5 gpurun_out/v_pytest.log | cut -c1-600
timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/v_train.json 2>> gpurun_out/v_bench.err
python -c "import json; d=json.load(open('gpurun_out/v_train.json')); print('TRAIN', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/v_bench.err
P3D_CONV_NO_CO64=1 P3D_WGRAD_NO_SMALL=1 timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/v_train_off.json 2>> gpurun_out/v_bench.err
python -c "import json; d=json.load(open('gpurun_out/v_train_off.json')); print('TRAIN no co64/small', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/v_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-exact-fp32 > gpurun_out/v_infer.json 2>> gpurun_out/v_bench.err
python -c "import json; d=json.load(open('gpurun_out/v_infer.json')); print('INFER', d['value'], d['ms_per_step'])" || tail -n 5 gpurun_out/v_bench.err
