#!/bin/bash
# round-6 session zf: the rest of session ze's parity cases (test sizes corrected) and the generator-level tests of the fused last layer
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zf
timeout 900 python -m pytest tests/test_split_acts.py -q -m gpu --tb=short -s > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -5 gpurun_out/${T}_gputest.log | cut -c1-400
grep "vs two launches" gpurun_out/${T}_gputest.log | cut -c1-200
python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; echo "trace exit $?"; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null
grep -n "r2_bf16x3\|torgb_wide\|one step" gpurun_out/${T}_step_trace.txt | cut -c1-160
echo finished
