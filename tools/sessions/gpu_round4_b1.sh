#!/bin/bash
# round 4, session b1: only the tests (each file in its own pytest process, log flushed per file, so that a lost box still tells which file it was in)
tag=${1:-round4_b1}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for f in test_discriminator test_train_step test_train_full test_dp_two_ranks_gpu test_loss_phases test_train_nograd_gpu; do
  echo "== $f" | tee -a gpurun_out/${tag}_tests.log
  timeout 600 python -m pytest tests/$f.py -m gpu -q -s --tb=short -rf -p no:cacheprovider >> gpurun_out/${tag}_tests.log 2>&1
  echo "rc=$?" | tee -a gpurun_out/${tag}_tests.log
done
grep -E "fp16-top-4|fp32 \{|conv-family|TWO_RANKS|worst parameters|passed|failed" gpurun_out/${tag}_tests.log | cut -c1-1000
grep -E "^E  |Error" gpurun_out/${tag}_tests.log | head -40 | cut -c1-1500
