#!/bin/bash
# round-5 session m: the bf16x6 formulation (P3D_F32_BF16X6) — parity of the new kernels, then the default bench line (its exact_fp32 object carries the
# backbone_as_bf16x6 sub-leg, train_step the fp32_as_bf16x6 variant), same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_m
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_conv_grad_gpu.py -q -m gpu -s -k "bf16x6" > gpurun_out/${T}_gputest.log 2>&1; echo "x6 tests exit $?"
grep -E "bf16x6|passed|failed" gpurun_out/${T}_gputest.log | cut -c1-260 | tail -40
timeout 900 python bench.py --no-configs > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/round5_m_bench_line.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'stage', d['stage_ms'])
e = d['exact_fp32']
print('exact', e.get('value'), e.get('stage_ms'), e.get('mfma_conv'))
print('x6', json.dumps(e.get('backbone_as_bf16x6'))[:900])
t = d['train_step']
print('train', t.get('ms_per_iteration'), t.get('phase_ms'))
print('train x6', json.dumps(t.get('fp32_as_bf16x6'))[:600])
print('train bf16x3', json.dumps(t.get('generator_bf16x3'))[:400])
PY
echo finished
