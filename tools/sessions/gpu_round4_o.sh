#!/bin/bash
# round 4, session o: the halo kernel's Co <= 64 wave mapping — the whole suite (every route of the kernel), training line, per-geometry table, inference line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/round4_o_gputest.log 2>&1; tail -n 5 gpurun_out/round4_o_gputest.log | cut -c1-400
grep -E "^E  " gpurun_out/round4_o_gputest.log | head -8 | cut -c1-1500
timeout 300 python bench.py --train-step --steps 3 --warmup 2 > gpurun_out/round4_o_train.json 2>> gpurun_out/round4_o_bench.err
python -c "import json; d=json.load(open('gpurun_out/round4_o_train.json')); print('TRAIN', d['ms_per_step'], d['train_step']['phase_ms'])" || tail -n 5 gpurun_out/round4_o_bench.err
timeout 600 python tests/gpu_time_train_convs.py > gpurun_out/round4_o_convs.log 2>&1; grep -E "  64   64 3 s1" gpurun_out/round4_o_convs.log | head -8 | cut -c1-160
timeout 300 python bench.py --no-train-step --no-cpu-baseline --no-exact-fp32 > gpurun_out/round4_o_bench_line_hipgraph.json 2>> gpurun_out/round4_o_bench.err
python -c "import json; d=json.load(open('gpurun_out/round4_o_bench_line_hipgraph.json')); print('INFER', d['value'], d['ms_per_step'], d['stage_ms'])"
