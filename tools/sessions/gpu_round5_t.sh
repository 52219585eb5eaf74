#!/bin/bash
# round-5 session t: two batches in flight (bench.py `pipelined`: a second capture of the step replayed on a second stream)
# (NOT KEPT: +0.7 % — profiles/round5_t_two_batches_in_flight_not_kept.log; the `pipelined` key was removed from bench.py again)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_t
timeout 400 python bench.py --no-cpu-baseline --no-train-step --no-exact-fp32 --no-configs > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err; echo "bench exit $?"
python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_line.json')); print(d['value'], d['ms_per_step'], d['stage_ms']); print(json.dumps(d['pipelined'])[:900])"
tail -3 gpurun_out/${T}_bench.err | cut -c1-300
echo finished
