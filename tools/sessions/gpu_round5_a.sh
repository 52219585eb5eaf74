#!/bin/bash
# round-5 session a: the MFMA-rate probe; the whole -m gpu suite (records the fp16 legs' measured errors); the default bench line with the new
# `configs` key; counter passes of the ray-marcher for this tree (render_device.h gained the debug bin-index record)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round5_a
timeout 300 python tools/probe_mfma_rate.py $T > gpurun_out/${T}_mfma_rate_probe.log 2>&1; echo "probe exit $?"
tail -4 gpurun_out/${T}_mfma_rate_probe.log | cut -c1-300
( timeout 900 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | tail -25 ) > gpurun_out/${T}_gputest.log 2>&1
tail -4 gpurun_out/${T}_gputest.log | cut -c1-600
cp gpurun_out/parity_errors.json gpurun_out/${T}_parity_errors.json 2>/dev/null
( timeout 600 python bench.py 2>gpurun_out/${T}_bench.err | tail -1 ) > gpurun_out/${T}_bench_line_default.json
cut -c1-400 gpurun_out/${T}_bench_line_default.json
timeout 500 python tests/gpu_pmc_render.py > gpurun_out/${T}_render_pmc.log 2>&1; echo "pmc exit $?"
P3D_MLP_BF16X3=0 timeout 500 python tests/gpu_pmc_render.py > gpurun_out/${T}_render_pmc_exact.log 2>&1; echo "pmc exact exit $?"
ls gpurun_out | head -40
