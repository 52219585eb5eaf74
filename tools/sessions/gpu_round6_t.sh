#!/bin/bash
# round-6 session t: the convolution suites with the bf16x6 kernel choice forced both ways (P3D_X6_PRESPLIT=2 inside the tests)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_t
timeout 1800 python -m pytest tests/test_conv_gpu.py tests/test_conv_grad_gpu.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -6 gpurun_out/${T}_gputest.log | cut -c1-300
echo finished
