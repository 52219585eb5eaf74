#!/bin/bash
# One GPU-box session: new-kernel tests first (verbose output kept), then the whole gpu suite, the training census and a bench line.
# usage: bash tools/sessions/gpu_session.sh <tag>
tag=${1:-s}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== new tests" > $out/summary.txt
timeout 1500 python -m pytest tests/test_conv_grad_gpu.py tests/test_model_full.py tests/test_train_step.py tests/test_discriminator.py -m gpu -q -s -p no:cacheprovider > $out/new_tests.log 2>&1
echo "new tests rc=$?" >> $out/summary.txt
tail -5 $out/new_tests.log >> $out/summary.txt
grep -E "FAILED|ERROR" $out/new_tests.log | head -60 >> $out/summary.txt
echo "== full gpu suite" >> $out/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_conv_grad_gpu.py --deselect tests/test_model_full.py > $out/gpu_suite.log 2>&1
echo "suite rc=$?" >> $out/summary.txt
tail -15 $out/gpu_suite.log >> $out/summary.txt
echo "== train census" >> $out/summary.txt
timeout 900 python tests/gpu_train_census.py 4 64 --json $out/train_census.json > $out/train_census.log 2>&1
echo "census rc=$?" >> $out/summary.txt
cat $out/train_census.log >> $out/summary.txt
echo "== bench" >> $out/summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/summary.txt
cat $out/bench.json >> $out/summary.txt
tail -3 $out/bench.err >> $out/summary.txt
cat $out/summary.txt
