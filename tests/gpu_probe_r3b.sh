#!/bin/bash
# round 3, session b: the tests that failed in session a, parity of the register-staged conv kernels, A/B timings (weights through registers vs
# LDS-DMA; first-round stagger), counter passes of the default ray-marcher with per-pass timeouts
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_render_bwd_gpu.py tests/test_train_step.py tests/test_train_full.py tests/test_discriminator.py tests/test_hazard_probe_gpu.py -m gpu -q --tb=short -rf -s -p no:cacheprovider > gpurun_out/r3b_tests.log 2>&1; tail -15 gpurun_out/r3b_tests.log
P3D_H2_REGW=1 P3D_UP2_REGW=1 timeout 300 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r3b_conv_tests.log 2>&1; tail -4 gpurun_out/r3b_conv_tests.log
for regw in 1 0; do for sl in 0 5; do
  echo "== up2 REGW=$regw STAGGER=$sl"; P3D_UP2_REGW=$regw P3D_STAGGER_SLEEPS=$sl timeout 120 python tests/gpu_probe_up2.py 2>&1 | grep "fused True"
done; done 2>&1 | tee gpurun_out/r3b_up2.log
for regw in 1 0; do for sl in 0 6; do
  echo "== h2 REGW=$regw STAGGER=$sl"; P3D_H2_REGW=$regw P3D_STAGGER_SLEEPS=$sl timeout 120 python tests/gpu_microbench_p3dconv.py 2>&1 | grep "p3d sr"
done; done 2>&1 | tee gpurun_out/r3b_h2.log
for regw in 1 0; do echo "== up2 K loop only REGW=$regw"; P3D_UP2_DEBUG=1 P3D_UP2_REGW=$regw timeout 120 python tests/gpu_probe_up2.py 2>&1 | grep "fused True"; done 2>&1 | tee -a gpurun_out/r3b_up2.log
timeout 700 python tests/gpu_pmc_render.py > gpurun_out/r3b_pmc.log 2>&1; tail -c 2500 gpurun_out/r3b_pmc.log
