#!/bin/bash
# round-4 session ah: the 64-channel form of the transposing-read fp16 weight gradient — parity, rows both ways
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/wgrad_variants.txt
timeout 600 python -m pytest tests/test_conv_grad_gpu.py tests/test_conv_layer_gpu.py -m gpu -q --tb=short > gpurun_out/ah_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/ah_pytest.log
tail -8 gpurun_out/ah_pytest.log | cut -c1-900
timeout 200 python tests/gpu_time_wgrad.py tr_small > /dev/null 2>gpurun_out/ah_err.log
P3D_WGRAD_NO_TR_SMALL=1 timeout 200 python tests/gpu_time_wgrad.py reg_small > /dev/null 2>>gpurun_out/ah_err.log
grep "64x64" gpurun_out/wgrad_variants.txt
