#!/bin/bash
# round-5 session b: parity of the tightened fp16 legs, of edge2car at configs[3] size, of the fused kernel's own bin indices and of the activation-free fused
# ToRGB; same-box A/B of the SR heads' dead-x skip (with / without the swapped-operand epilogue); kernel statistics + counter passes of the edge2car renderer
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_b
( timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_model_full.py tests/test_model_gpu.py tests/test_srheads.py tests/test_render_gpu.py tests/test_hazard_probe_gpu.py tests/test_bench_two_ranks_gpu.py tests/test_dropin.py -q -m gpu --tb=short 2>&1 | tail -30 ) > gpurun_out/${T}_gputest.log 2>&1
tail -5 gpurun_out/${T}_gputest.log | cut -c1-700
cp gpurun_out/parity_errors.json gpurun_out/${T}_parity_errors.json 2>/dev/null
B="python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs --steps 40"
for rep in 1 2; do
  ( P3D_SR_SKIP_DEAD_X=0 timeout 300 $B 2>/dev/null | tail -1 ) > gpurun_out/${T}_ab_store_x_$rep.json
  ( P3D_TORGB_NO_TR=1 timeout 300 $B 2>/dev/null | tail -1 ) > gpurun_out/${T}_ab_skip_lds_epilogue_$rep.json
  ( timeout 300 $B 2>/dev/null | tail -1 ) > gpurun_out/${T}_ab_skip_tr_$rep.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/round5_b_ab_*.json')):
    try:
        d = json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['stage_ms'], d['mfma_conv']['conv_f16'])
    except Exception as e: print(f, 'ERR', e)
PY
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e2c -o e -- python $GRAFT_REPO_ROOT/bench.py --dataset edge2car --batch 8 --steps 20 --warmup 3 --no-cpu-baseline --no-graph --no-train-step --no-exact-fp32 --no-configs 2>/dev/null | tail -1 ) > gpurun_out/${T}_edge2car_bench_eager.json
find /tmp/prof_e2c -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_edge2car_kernel_stats.csv \;
head -12 gpurun_out/${T}_edge2car_kernel_stats.csv | cut -c1-200
P3D_PMC_DATASET=edge2car P3D_PMC_GROUPS=0,1,2,3,4,5 timeout 400 python tests/gpu_pmc_render.py > gpurun_out/${T}_render_pmc_edge2car.log 2>&1; echo "pmc edge2car exit $?"
tail -2 gpurun_out/${T}_render_pmc_edge2car.log | cut -c1-1500
