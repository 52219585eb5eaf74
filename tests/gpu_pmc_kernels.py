"""SQ counter passes over EVERY kernel of a workload, aggregated per kernel name: which resource each of them keeps busiest.

    python tests/gpu_pmc_kernels.py infer     rocprofv3 --pmc passes over `bench.py --no-graph ... --steps 4` (the inference step, eager launches)
    python tests/gpu_pmc_kernels.py train     ... over tests/gpu_train_census.py 4 128 --no-census (one G pass + one D pass, config 3)
    python tests/gpu_pmc_kernels.py train6    ... over `bench.py --train-step --steps 1 --warmup 1` (the six-phase iteration of config 3), plus FETCH_SIZE / WRITE_SIZE
                                              passes: memory-side GB/s per kernel (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md)

Three passes (kernel-trace + one counter group each, no other trace domain; each under its own timeout).  Writes gpurun_out/kernel_pmc_<what>.json
and .txt: per kernel (sorted by total time) launches, average duration, and over the kernel's life: VALU issue, matrix pipe, LDS array
utilisation, waves per SIMD, and how the waves' time splits into issuing / parked at s_waitcnt / issue-stalled.
Units as in tests/gpu_pmc_render.py (SQ_* quad-cycles summed over waves, MFMA busy cycles summed over SIMDs, GRBM over 8 XCDs)."""
import collections
import csv
import glob
import json
import os
import signal
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD, N_CU, N_XCD = 1024, 256, 8
GROUPS = [
    ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'GRBM_GUI_ACTIVE'],
    ['SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SALU', 'SQ_LDS_IDX_ACTIVE'],
    ['SQ_LDS_BANK_CONFLICT', 'SQ_WAIT_INST_LDS', 'SQ_INST_LEVEL_VMEM', 'SQ_INST_LEVEL_LDS', 'SQ_WAVES', 'SQ_BUSY_CYCLES'],
]
TRAFFIC_GROUPS = [['FETCH_SIZE'], ['WRITE_SIZE']]
HBM_PEAK_GBS = 8000.0
WORKLOADS = {
    'train6': [sys.executable, os.path.join(ROOT, 'bench.py'), '--train-step', '--steps', '1', '--warmup', '1'],
    'infer': [sys.executable, os.path.join(ROOT, 'bench.py'), '--no-graph', '--no-cpu-baseline', '--no-train-step', '--no-exact-fp32', '--no-configs', '--steps', '4', '--warmup', '2'],
    'train': [sys.executable, os.path.join(ROOT, 'tests', 'gpu_train_census.py'), '4', '128', '--no-census'],
}
PASS_TIMEOUT_S = int(os.environ.get('PMC_PASS_TIMEOUT', 150))


def short(name):
    name = name.replace('void ', '').replace('p3d::', '')
    return name.split('(')[0][:70]


def one_pass(what, idx, counters, out_root):
    out_dir = os.path.join(out_root, f'{what}_pass{idx}')
    os.makedirs(out_dir, exist_ok=True)
    cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + counters + ['--output-format', 'csv', '-d', out_dir, '-o', 'r', '--'] + WORKLOADS[what]
    proc = subprocess.Popen(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=PASS_TIMEOUT_S)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        proc.communicate()
        return None, None, f'pass timed out after {PASS_TIMEOUT_S} s'
    vals, dur = collections.defaultdict(lambda: collections.defaultdict(list)), collections.defaultdict(list)
    for f in glob.glob(os.path.join(out_dir, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                vals[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    for f in glob.glob(os.path.join(out_dir, '**', '*kernel_trace.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                dur[short(row['Kernel_Name'])].append((float(row['End_Timestamp']) - float(row['Start_Timestamp'])) * 1e-3)
    if not vals:
        return None, None, out[-1200:]
    return vals, dur, None


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'infer'
    out_root = os.path.join(ROOT, 'gpurun_out', 'kpmc')
    per, durs, errors = collections.defaultdict(dict), {}, {}
    groups = GROUPS + (TRAFFIC_GROUPS if what == 'train6' else [])
    for i, grp in enumerate(groups):
        vals, dur, err = one_pass(what, i, grp, out_root)
        if vals is None:
            errors[' '.join(grp)] = err
            continue
        for k, cs in vals.items():
            for c, v in cs.items():
                per[k][c] = sum(v) / len(v)                       # average per launch of that kernel
        if grp in GROUPS:
            durs = dur                                            # durations of an SQ pass (the traffic passes may serialise differently)
    rows = []
    for k, c in per.items():
        d = durs.get(k, [])
        if not d or 'GRBM_GUI_ACTIVE' not in c:
            continue
        cyc = c['GRBM_GUI_ACTIVE'] / N_XCD
        if cyc <= 0:
            continue
        w = c.get('SQ_WAVE_CYCLES', 0.0)
        rows.append({
            'kernel': k, 'launches': len(d), 'avg_us': sum(d) / len(d), 'total_ms': sum(d) / 1e3,
            'valu_issue': c.get('SQ_ACTIVE_INST_VALU', 0) * 4 / (N_SIMD * cyc),
            'mfma_pipe': c['SQ_VALU_MFMA_BUSY_CYCLES'] / (N_SIMD * cyc) if 'SQ_VALU_MFMA_BUSY_CYCLES' in c else None,
            'lds_array': c['SQ_LDS_IDX_ACTIVE'] / (N_CU * cyc) if 'SQ_LDS_IDX_ACTIVE' in c else None,
            'waves_per_simd': w * 4 / (N_SIMD * cyc),
            'issuing': c.get('SQ_ACTIVE_INST_ANY', 0) / w if w else None, 'parked_at_waitcnt': c.get('SQ_WAIT_ANY', 0) / w if w else None,
            'issue_stalled': c.get('SQ_WAIT_INST_ANY', 0) / w if w else None,
            'lds_bank_conflict_share': c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'] if c.get('SQ_LDS_IDX_ACTIVE') and 'SQ_LDS_BANK_CONFLICT' in c else None,
            'insts_per_launch': {n: c[n] for n in ('SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SALU') if n in c},
            'hbm_bytes_per_launch': (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c else None,
        })
        r = rows[-1]
        r['hbm_frac'] = r['hbm_bytes_per_launch'] / (r['avg_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS if r['hbm_bytes_per_launch'] is not None else None
    rows.sort(key=lambda r: -r['total_ms'])
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump({'workload': ' '.join(WORKLOADS[what][1:]), 'kernels': rows, 'errors': errors}, open(os.path.join(ROOT, 'gpurun_out', f'kernel_pmc_{what}.json'), 'w'), indent=1)
    f2 = lambda v: '   - ' if v is None else f'{v:5.2f}'
    with open(os.path.join(ROOT, 'gpurun_out', f'kernel_pmc_{what}.txt'), 'w') as f:
        f.write(f'# {what}: {" ".join(WORKLOADS[what][1:])}\n# utilisation over each kernel\'s life (1.0 = every SIMD / CU busy every cycle); wave time split sums to ~1\n')
        f.write(f'{"kernel":70s} {"n":>5s} {"avg us":>9s} {"tot ms":>8s}  valu  mfma   lds  w/simd | issue  wait stall |  hbm\n')
        for r in rows[:40]:
            f.write(f"{r['kernel']:70s} {r['launches']:5d} {r['avg_us']:9.1f} {r['total_ms']:8.2f} {f2(r['valu_issue'])} {f2(r['mfma_pipe'])} {f2(r['lds_array'])} {f2(r['waves_per_simd'])}  | "
                    f"{f2(r['issuing'])} {f2(r['parked_at_waitcnt'])} {f2(r['issue_stalled'])} | {f2(r.get('hbm_frac'))}\n")
        for k, v in errors.items():
            f.write(f'# FAILED PASS [{k}]: {v[-300:]}\n')
    print(open(os.path.join(ROOT, 'gpurun_out', f'kernel_pmc_{what}.txt')).read())
    import shutil
    shutil.rmtree(out_root, ignore_errors=True)              # raw per-pass CSVs: tens of MB for a training iteration; gpurun copies back at most 64 MiB


if __name__ == '__main__':
    main()
