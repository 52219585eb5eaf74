#!/bin/bash
# round-4 session aa: is the training iteration GPU-bound throughout?  kernel trace of three iterations -> busy time, idle gaps by size, the kernels that precede the long gaps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; mkdir -p gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gap -o t -- python $GRAFT_REPO_ROOT/bench.py --train-step --steps 3 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/aa_train_profiled.json 2> /tmp/prof_gap.err)
f=$(find /tmp/prof_gap -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' > gpurun_out/aa_gaps.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
# the last 3 iterations: find them by the G_ema foreach lerp? simpler: take the last 60 % of the trace by time
t0, t1 = ev[0][0], ev[-1][1]
cut = t1 - (t1 - t0) * 0.45
ev = [e for e in ev if e[0] >= cut]
span = ev[-1][1] - ev[0][0]
busy = 0; cur_end = ev[0][0]; gaps = []
for s, e, nme in ev:
    if s > cur_end:
        gaps.append((s - cur_end, prev_name, nme))
    busy += max(0, e - max(s, cur_end)); 
    if e > cur_end: cur_end = e; prev_name = nme
print(f'window {span/1e6:.1f} ms, {len(ev)} kernels, GPU busy {busy/1e6:.1f} ms = {busy/span:.3f}, idle {(span-busy)/1e6:.1f} ms')
hist = collections.Counter()
for g, _, _ in gaps:
    b = '<2us' if g < 2000 else '2-5us' if g < 5000 else '5-10us' if g < 10000 else '10-20us' if g < 20000 else '20-50us' if g < 50000 else '50-200us' if g < 200000 else '>200us'
    hist[b] += g
for b in ('<2us', '2-5us', '5-10us', '10-20us', '20-50us', '50-200us', '>200us'):
    print(f'  gaps {b:9s}: {hist[b]/1e6:8.2f} ms total')
after = collections.Counter(); cnt = collections.Counter()
for g, p, n in gaps:
    if g >= 5000:
        after[n[:70]] += g; cnt[n[:70]] += 1
print('kernels that START after a gap >= 5 us (waiting for the host), by total gap:')
for k, v in after.most_common(25):
    print(f'  {v/1e6:7.2f} ms  x{cnt[k]:5d}  {k}')
PY
head -40 gpurun_out/aa_gaps.txt
