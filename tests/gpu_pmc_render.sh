# SQ counter passes over the fused ray-marcher (tests/gpu_profile_render.py); one counter group per pass, kernel-trace only.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/rpmc
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && REPS=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/rp$i -o r -- python $GRAFT_REPO_ROOT/tests/gpu_profile_render.py > /tmp/rp$i.log 2>&1 )
  f=$(find /tmp/rp$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then grep render_forward "$f" | python3 -c "
import sys, csv, collections
acc = collections.defaultdict(list)
for row in csv.reader(sys.stdin):
    acc[row[15]].append(float(row[16]))
for k, v in acc.items(): print(k, len(v), sum(v) / len(v))
" > gpurun_out/rpmc/pass$i.txt; else tail -5 /tmp/rp$i.log > gpurun_out/rpmc/pass$i.txt; fi
  cat gpurun_out/rpmc/pass$i.txt
done
