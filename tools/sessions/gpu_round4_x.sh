#!/bin/bash
# round-4 session x: census of the ATen glue launches of the training iteration by Python line (three frames deep)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
P3D_CENSUS_DEPTH=3 timeout 600 python tests/gpu_aten_census.py > gpurun_out/x_census.log 2>&1; tail -2 gpurun_out/x_census.log | cut -c1-300
