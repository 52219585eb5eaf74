#!/bin/bash
# round-5 session v: the default bench line once more, now that profiles/kernel_pmc_train6.json (which train_step.roofline quotes) is the final tree's
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_v
( timeout 900 python bench.py 2>gpurun_out/${T}_bench.err | tail -1 ) > gpurun_out/${T}_bench_line_default.json
cut -c1-200 gpurun_out/${T}_bench_line_default.json
echo finished
