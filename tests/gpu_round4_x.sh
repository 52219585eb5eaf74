#!/bin/bash
# round-4 session x: census of the ATen glue launches of the training iteration by Python line; the optimizer steps after the stride fix
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python tests/gpu_aten_census.py > gpurun_out/x_census.log 2>&1; tail -3 gpurun_out/x_census.log | cut -c1-300
timeout 500 python tests/gpu_probe_adam.py 2>&1 | grep -v amdgpu.ids | grep -A2 "^Gmain" | cut -c1-600
