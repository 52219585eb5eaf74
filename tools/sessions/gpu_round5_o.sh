#!/bin/bash
# round-5 session o: the whole GPU suite with P3D_F32_BF16X6=1 (every fp32 convolution that would take the f32-input MFMA formed as six bf16 MFMAs per product) —
# is the arithmetic a drop-in for the exact kernels under the suite's own bounds?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_o
P3D_F32_BF16X6=1 timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/${T}_gputest_x6.log 2>&1; echo "suite exit $?"
tail -25 gpurun_out/${T}_gputest_x6.log | cut -c1-300
cp gpurun_out/parity_errors.json gpurun_out/${T}_parity_errors_x6.json 2>/dev/null
echo finished
