#!/bin/bash
# round-6 session zg: the TR epilogue with everything staged in LDS (no global round trips in its loops): parity, the launch in the step trace, A/B of the line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zg
timeout 900 python -m pytest tests/test_split_acts.py -q -m gpu --tb=short -s > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -3 gpurun_out/${T}_gputest.log | cut -c1-400
grep "vs two launches" gpurun_out/${T}_gputest.log | tail -7 | cut -c1-200
python tests/gpu_step_trace.py > gpurun_out/${T}_step_trace.log 2>&1; echo "trace exit $?"; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null
grep -n "r2_bf16x3\|torgb_wide\|one step" gpurun_out/${T}_step_trace.txt | cut -c1-160
for rep in 1 2 3; do
  for v in 0 1; do
    P3D_FUSE_CONV_WIDE_TORGB=$v timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs --no-exact-fp32 > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('fused=$v rep $rep:', d['value'], 'img/s,', d['ms_per_step'], 'ms,', d.get('stage_ms'), 'conv_bf16x3', d.get('mfma_conv', {}).get('conv_bf16x3'))" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
echo finished
