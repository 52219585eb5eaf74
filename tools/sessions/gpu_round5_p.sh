#!/bin/bash
# round-5 session p: the weight gradient as bf16x6 (conv_wgrad_kernel<float, 16, true, false, true>) — parity, the kernel alone on the iteration's geometries
# (exact vs bf16x6, interleaved), then the training bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_p
timeout 900 python -m pytest tests/test_conv_grad_gpu.py -q -m gpu -s -k "bf16x6" > gpurun_out/${T}_gputest.log 2>&1; echo "x6 tests exit $?"
grep -E "passed|failed|Error|assert" gpurun_out/${T}_gputest.log | cut -c1-300 | tail -12
rm -f gpurun_out/wgrad_variants.txt
for rep in 1 2; do
  WGRAD_X6=0 timeout 200 python tests/gpu_time_wgrad.py exact > /dev/null 2>&1
  WGRAD_X6=1 timeout 200 python tests/gpu_time_wgrad.py bf16x6 > /dev/null 2>&1
done
grep float32 gpurun_out/wgrad_variants.txt | sort -k2,6 -s | cut -c1-120
cp gpurun_out/wgrad_variants.txt gpurun_out/${T}_wgrad_variants.txt
timeout 900 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/${T}_bench_line_train.json 2> gpurun_out/${T}_bench_train.err; echo "train bench exit $?"
P3D_WGRAD_BF16X6=0 timeout 900 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/${T}_bench_line_train_wgrad_exact.json 2>> gpurun_out/${T}_bench_train.err; echo "train bench (exact wgrad) exit $?"
python - <<'PY'
import json
for f in ('round5_p_bench_line_train.json', 'round5_p_bench_line_train_wgrad_exact.json'):
    d = json.load(open('gpurun_out/' + f))
    t = d['train_step']
    print(f, t['ms_per_iteration'], t['phase_ms'], 'lazy', t['lazy_schedule']['ms_per_iteration'])
PY
echo finished
