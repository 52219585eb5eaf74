#!/bin/bash
# round-5 session c: the model-size parity tests with the tile-sum bound, ATen census of the inference step and of the training iteration (this tree),
# the --train-step line (arithmetic floor per phase)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_c
( timeout 600 python -m pytest tests/test_model_full.py tests/test_conv_gpu.py -q -m gpu --tb=short 2>&1 | tail -12 ) > gpurun_out/${T}_gputest.log 2>&1
tail -3 gpurun_out/${T}_gputest.log | cut -c1-500
timeout 200 python tools/gpu_infer_aten_census.py > gpurun_out/${T}_infer_census.log 2>&1; echo "infer census exit $?"
timeout 400 python tests/gpu_aten_census.py > gpurun_out/${T}_train_census.log 2>&1; echo "train census exit $?"
cp gpurun_out/aten_census.txt gpurun_out/${T}_train_aten_census.txt 2>/dev/null
( timeout 600 python bench.py --train-step --steps 3 --warmup 2 2>gpurun_out/${T}_train.err | tail -1 ) > gpurun_out/${T}_bench_line_train.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/round5_c_bench_line_train.json'))
t = d['train_step']; print(t['ms_per_iteration'], t['phase_ms'])
print(json.dumps(t.get('arithmetic_floor'))[:3000])
PY
head -50 gpurun_out/infer_aten_census.txt | cut -c1-330
