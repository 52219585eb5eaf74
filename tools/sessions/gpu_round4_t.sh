#!/bin/bash
# round-4 session t: the weight-gradient kernel's lean loader (FAST), its split plan (all work-groups resident, equal share per XCD) and the half-tile pixel split
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/wgrad_variants.txt
timeout 600 python -m pytest tests/test_conv_grad_gpu.py -m gpu -x -q > gpurun_out/t_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/t_pytest.log
tail -3 gpurun_out/t_pytest.log
timeout 200 python tests/gpu_time_wgrad.py xcd_r4 > /dev/null 2>gpurun_out/t_err.log
P3D_WGRAD_NO_HALF=1 timeout 200 python tests/gpu_time_wgrad.py xcd_r4_nohalf > /dev/null 2>>gpurun_out/t_err.log
P3D_WGRAD_WG_PER_CU=3 timeout 200 python tests/gpu_time_wgrad.py xcd_r3 > /dev/null 2>>gpurun_out/t_err.log
P3D_WGRAD_PLAN_OLD=1 P3D_WGRAD_NO_HALF=1 timeout 200 python tests/gpu_time_wgrad.py old_plan > /dev/null 2>>gpurun_out/t_err.log
cat gpurun_out/wgrad_variants.txt; grep -v amdgpu.ids gpurun_out/t_err.log | tail -5
timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/t_train.json 2>> gpurun_out/t_bench.err
python -c "import json; d=json.load(open('gpurun_out/t_train.json')); print('TRAIN', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/t_bench.err
P3D_WGRAD_PLAN_OLD=1 P3D_WGRAD_NO_HALF=1 P3D_WGRAD_NO_FAST=1 timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/t_train_old.json 2>> gpurun_out/t_bench.err
python -c "import json; d=json.load(open('gpurun_out/t_train_old.json')); print('TRAIN old', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/t_bench.err
