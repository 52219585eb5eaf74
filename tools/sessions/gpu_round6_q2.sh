cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python tests/gpu_time_train_convs.py > /dev/null 2>gpurun_out/round6_q_convs.err; cp gpurun_out/train_conv_geometries.txt gpurun_out/round6_q_train_conv_geometries.txt; tail -3 gpurun_out/round6_q_convs.err
