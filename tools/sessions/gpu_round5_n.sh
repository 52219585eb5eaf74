#!/bin/bash
# round-5 session n: the in-register splits written on pairs (conv2d.hip split_bf16x8 / split3_bf16x8, up2_fir.hip ub_split) — every convolution test, the
# ray-marcher with split8 on pairs as an ablation build (tools/build_render_variants.py 0 1024), then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_n
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_grad_gpu.py tests/test_split_acts.py tests/test_srheads.py -q -m gpu -x > gpurun_out/${T}_gputest.log 2>&1; echo "conv tests exit $?"
tail -3 gpurun_out/${T}_gputest.log
: > gpurun_out/${T}_render_variants.log
for rep in 1 2 3; do
  for bits in 0 1024; do
    echo "round $rep variant $bits: $(P3D_LIB_PATH=$GRAFT_REPO_ROOT/pix2pix3d_amd/libp3d_hip_rv$bits.so ITERS=20 timeout 120 python tests/gpu_profile_render.py 2>/dev/null | tail -1 | cut -c1-60)" >> gpurun_out/${T}_render_variants.log
  done
done
cat gpurun_out/${T}_render_variants.log
timeout 900 python bench.py --no-configs > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/round5_n_bench_line.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'stage', d['stage_ms'], d['mfma_conv'])
e = d['exact_fp32']
print('exact', e.get('value'), e.get('stage_ms'), e.get('mfma_conv'))
print('x6', json.dumps(e.get('backbone_as_bf16x6'))[:900])
t = d['train_step']
print('train', t.get('ms_per_iteration'), t.get('phase_ms'))
print('train x6', json.dumps(t.get('fp32_as_bf16x6'))[-330:])
print('train bf16x3', json.dumps(t.get('generator_bf16x3'))[-330:])
PY
echo finished
