#!/bin/bash
# round-6 session j: wide stores in conv3x3_halo_kernel's epilogue (fp32: 16 bytes, fp16: 8 bytes; P3D_CONV_STORE4=1 = one value per store, which now switches the
# generic kernel's AND the halo kernel's epilogue back) — parity of the training-side suites, then the training iteration both ways, interleaved, with the kernel time
# of each variant summed by rocprofv3 (the iteration is the sum of its kernels only where the device is the bottleneck)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_j
timeout 1800 python -m pytest tests/test_conv_gpu.py tests/test_conv_grad_gpu.py tests/test_conv_layer_gpu.py tests/test_discriminator.py tests/test_train_full.py tests/test_loss_phases.py tests/test_train_step.py tests/test_model_gpu.py -q -m gpu -x --tb=short > gpurun_out/${T}_gputest.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/${T}_gputest.log | cut -c1-300
for rep in 1 2 3; do
  for v in 1 0; do
    P3D_CONV_STORE4=$v timeout 600 python bench.py --train-step --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_train_${v}_${rep}.json 2>gpurun_out/${T}_train_${v}_${rep}.err
    python -c "
import json; d = json.load(open('gpurun_out/${T}_train_${v}_${rep}.json')); t = d.get('train_step', d); print('conv_store4=$v rep $rep', d.get('ms_per_step'), t.get('phase_ms'))" || tail -5 gpurun_out/${T}_train_${v}_${rep}.err
  done
done
export TMPDIR=/tmp
for v in 1 0; do
  ( cd /tmp && P3D_CONV_STORE4=$v timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t$v -o e -- python $GRAFT_REPO_ROOT/bench.py --train-step --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 ) > /dev/null
  find /tmp/prof_t$v -name '*kernel_stats.csv' -exec cp {} gpurun_out/${T}_train_kernel_stats_${v}.csv \;
  python - <<EOF
import csv
rows=list(csv.DictReader(open('gpurun_out/${T}_train_kernel_stats_${v}.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6
pick=lambda s: sum(float(r['TotalDurationNs']) for r in rows if s in r['Name'])/1e6
print('store4=$v total kernel ms', round(tot,1), 'halo', round(pick('conv3x3_halo_kernel'),1), 'generic', round(pick('conv2d_nhwc_kernel'),1), 'wgrad', round(pick('conv_wgrad'),1))
EOF
done
echo finished
