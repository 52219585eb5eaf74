# SQ counter passes over the renderer's backward kernel (tests/gpu_time_render_train.py 4 128, fused policy only); one group per pass.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/rbpmc
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && P3D_ONLY_FUSED=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/rbp$i -o r -- python $GRAFT_REPO_ROOT/tests/gpu_time_render_train.py 4 128 > /tmp/rbp$i.log 2>&1 )
  f=$(find /tmp/rbp$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then grep render_backward "$f" | python3 -c "
import sys, csv, collections
acc = collections.defaultdict(list)
for row in csv.reader(sys.stdin):
    acc[row[15]].append(float(row[16]))
for k, v in acc.items(): print(k, len(v), sum(v) / len(v))
" > gpurun_out/rbpmc/pass$i.txt; else tail -5 /tmp/rbp$i.log > gpurun_out/rbpmc/pass$i.txt; fi
  cat gpurun_out/rbpmc/pass$i.txt
done
