#!/bin/bash
# round-6 session c: (1) the plan's waits — a stream's second wait on a plan is for everything issued so far, finish_prefetch adds no edge once joined
# (P3D_PLAN_WAIT_LATEST) — with the heads' plans issued ahead; (2) fused Adam in the training iteration (P3D_BENCH_FUSED_ADAM).  Parity first.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_c
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_model_full.py tests/test_small_ops_gpu.py tests/test_srheads.py tests/test_model_variants.py tests/test_checkpoint.py tests/test_train_step.py tests/test_train_nograd_gpu.py tests/test_bench_two_ranks_gpu.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2 3; do
  for v in 00 01 11; do
    P3D_PLAN_WAIT_LATEST=${v:0:1} P3D_SR_PREFETCH_AHEAD=${v:1:1} timeout 300 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_${v}_${rep}.json 2>gpurun_out/${T}_bench_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_${v}_${rep}.json')); print('latest,ahead=$v rep $rep', d['value'], d['ms_per_step'], d['stage_ms'])" || tail -5 gpurun_out/${T}_bench_${v}_${rep}.err
  done
done
for rep in 1 2; do
  for v in 0 1; do
    P3D_BENCH_FUSED_ADAM=$v timeout 600 python bench.py --train-step --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_train_fused${v}_${rep}.json 2>gpurun_out/${T}_train_fused${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_train_fused${v}_${rep}.json')); t = d.get('train_step', d); print('fused_adam=$v rep $rep', d.get('ms_per_step'), t.get('phase_ms'))" || tail -5 gpurun_out/${T}_train_fused${v}_${rep}.err
  done
done
echo finished
