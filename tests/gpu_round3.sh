#!/bin/bash
# One GPU session of round 3: the -m gpu suite, the counter passes of the ray-marcher (default + exact-fp32 decoder), the benchmark line with
# the fresh counters in place, and the rocprofv3 kernel statistics.   usage: bash tests/gpu_round3.sh <tag> [quick]
tag=${1:-r3}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/${tag}_gputest.log 2>&1
tail -25 gpurun_out/${tag}_gputest.log
timeout 600 python tests/gpu_pmc_render.py > gpurun_out/${tag}_pmc.log 2>&1 && cp gpurun_out/render_pmc.json profiles/render_pmc.json
P3D_MLP_BF16X3=0 timeout 600 python tests/gpu_pmc_render.py > gpurun_out/${tag}_pmc_exact.log 2>&1 && cp gpurun_out/render_pmc_exact_fp32.json profiles/render_pmc_exact_fp32.json
tail -c 1500 gpurun_out/${tag}_pmc.log
timeout 900 python bench.py > gpurun_out/${tag}_bench_line_default.json 2> gpurun_out/${tag}_bench.err
head -c 3000 gpurun_out/${tag}_bench_line_default.json; tail -5 gpurun_out/${tag}_bench.err
if [ "$2" != "quick" ]; then timeout 900 bash tests/gpu_profiles.sh ${tag}; fi
