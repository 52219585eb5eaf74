#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_two_ranks_gpu.py tests/test_dp_two_ranks_gpu.py -m gpu -q -s --tb=short -rf -p no:cacheprovider > gpurun_out/round4_q_tests.log 2>&1; tail -n 40 gpurun_out/round4_q_tests.log | cut -c1-600
