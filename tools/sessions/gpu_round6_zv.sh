#!/bin/bash
# round-6 session zv: conv2d_nhwc_kernel<.., ISC> requests its scale table with the first tile (one rendezvous instead of two round trips in a row): parity, smoke, the launches in a step trace, the line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zv
( timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_layer_gpu.py tests/test_model_gpu.py tests/test_model_full.py -q -m gpu -x --tb=short 2>&1 | tail -5 ) > gpurun_out/${T}_gputest.log 2>&1
grep -n "passed\|failed" gpurun_out/${T}_gputest.log | tail -2
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2 ) > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
python tests/gpu_step_trace.py > /dev/null 2>&1; cp gpurun_out/step_trace.txt gpurun_out/${T}_step_trace.txt 2>/dev/null
head -1 gpurun_out/${T}_step_trace.txt | cut -c1-90; grep "true, false, false, true, false" gpurun_out/${T}_step_trace.txt | cut -c1-110
timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-configs --no-exact-fp32 2>/dev/null | python -c "
import json,sys; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('line:', d['value'], 'img/s,', d['ms_per_step'], 'ms,', d.get('stage_ms'))"
echo finished
