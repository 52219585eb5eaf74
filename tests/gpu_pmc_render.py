"""Counter passes over the fused ray-marcher at the bench workload: which resource binds it, and the memory-side traffic.

One `rocprofv3 --kernel-trace --pmc <group>` pass per counter group over tests/gpu_profile_render.py (no other trace domain; FETCH_SIZE and
WRITE_SIZE in passes of their own, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Writes

    gpurun_out/render_pmc.json      (render_pmc_exact_fp32.json under P3D_MLP_BF16X3=0) every counter (average per launch of render_forward_kernel), the kernel's duration in each pass, the
                                    derived utilisation of every candidate resource, the binding one, the traffic figure and the SHA-256 of
                                    the kernel sources — bench.py quotes `roofline.traffic` / `roofline.binding` from the committed copy
                                    (profiles/render_pmc.json) only while that hash matches the tree it benchmarks
    gpurun_out/render_sq_pmc.txt    the same as a table

    python tests/gpu_pmc_render.py            (on the GPU box)

Units (MI355X_MICROARCH.md, PMC section): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; *_sum counters over their instances."""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ['pix2pix3d_amd/csrc/render.hip', 'pix2pix3d_amd/csrc/render_device.h']
KERNEL = 'render_forward_kernel'
N_SIMD, N_CU, N_XCD = 1024, 256, 8
PEAK_CLOCK_HZ = 2.4e9                       # MI355X_MICROARCH.md: max clock
L2_PEAK_GBS, HBM_PEAK_GBS = 34500.0, 8000.0
LINE = 128                                  # bytes per L1 / L2 line on gfx950

GROUPS = [          # the SQ / GRBM and the two traffic passes first (known to run in seconds); the texture-path blocks last, few counters per pass, each pass
    ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM'],   # under its own
    ['SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_SALU', 'SQ_ACTIVE_INST_SCA', 'SQ_INSTS_VALU_TRANS_F32'],   # timeout
    ['SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_ADDR_CONFLICT', 'SQ_WAIT_INST_LDS', 'SQ_INST_LEVEL_VMEM', 'SQ_INST_LEVEL_LDS', 'SQ_WAVES', 'GRBM_GUI_ACTIVE'],
    ['FETCH_SIZE'],
    ['WRITE_SIZE'],
    ['TCC_HIT_sum', 'TCC_MISS_sum'],
    ['TCC_REQ_sum', 'TCC_READ_sum'],
    ['TCP_TOTAL_CACHE_ACCESSES_sum', 'TCP_TCC_READ_REQ_sum'],
    ['TA_TA_BUSY_sum', 'TA_BUFFER_READ_WAVEFRONTS_sum'],
]
PASS_TIMEOUT_S = int(os.environ.get('PMC_PASS_TIMEOUT', 75))       # (round 3: a four-counter TA pass never returned and cost its whole 600 s)


def kernel_source_hash(root=ROOT):
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(root, rel), 'rb').read())
    return h.hexdigest()[:16]


def one_pass(idx, counters, out_root):
    out_dir = os.path.join(out_root, f'pass{idx}')
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, REPS='1', TMPDIR='/tmp')
    cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + counters + ['--output-format', 'csv', '-d', out_dir, '-o', 'r', '--',
                                                                   sys.executable, os.path.join(ROOT, 'tests', 'gpu_profile_render.py')]
    import signal
    proc = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=PASS_TIMEOUT_S)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)                      # the pass's own process group (rocprofv3 + the python child), nothing else
        proc.communicate()
        return None, None, f'pass timed out after {PASS_TIMEOUT_S} s'
    vals, dur = {}, []
    for f in glob.glob(os.path.join(out_dir, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if KERNEL in row.get('Kernel_Name', ''):
                    vals.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
    for f in glob.glob(os.path.join(out_dir, '**', '*kernel_trace.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if KERNEL in row.get('Kernel_Name', ''):
                    dur.append((float(row['End_Timestamp']) - float(row['Start_Timestamp'])) * 1e-6)
    if not vals:
        return None, None, out[-1500:]
    return {k: sum(v) / len(v) for k, v in vals.items()}, (sum(dur) / len(dur) if dur else None), None


def derive(c, ms):
    """Utilisation of every candidate resource over the kernel's life (cycles per XCD = GRBM_GUI_ACTIVE / 8)."""
    d = {}
    cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / N_XCD
    if cyc <= 0:
        return d, None
    d['cycles_per_launch'] = cyc
    d['effective_clock_GHz'] = cyc / (ms['GRBM_GUI_ACTIVE'] * 1e-3) / 1e9 if ms.get('GRBM_GUI_ACTIVE') else None
    simd_cyc, cu_cyc = N_SIMD * cyc, N_CU * cyc
    util = {}
    if 'SQ_ACTIVE_INST_VALU' in c:
        util['valu_issue'] = c['SQ_ACTIVE_INST_VALU'] * 4 / simd_cyc
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        util['mfma_pipe'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cyc
    if 'SQ_LDS_IDX_ACTIVE' in c:
        util['lds_array'] = c['SQ_LDS_IDX_ACTIVE'] / cu_cyc
    if 'TA_TA_BUSY_sum' in c:
        util['texture_addresser'] = c['TA_TA_BUSY_sum'] / cu_cyc
    if 'TCP_TCC_READ_REQ_sum' in c and ms.get('TCP_TCC_READ_REQ_sum'):
        util['l2_bandwidth'] = c['TCP_TCC_READ_REQ_sum'] * LINE / (ms['TCP_TCC_READ_REQ_sum'] * 1e-3) / 1e9 / L2_PEAK_GBS
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c and ms.get('FETCH_SIZE'):
        util['hbm_bandwidth'] = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / (ms['FETCH_SIZE'] * 1e-3) / 1e9 / HBM_PEAK_GBS
    d['utilisation'] = util
    if 'SQ_WAVE_CYCLES' in c:
        w = c['SQ_WAVE_CYCLES']
        d['wave_time_split'] = {k: c[n] / w for k, n in (('issuing', 'SQ_ACTIVE_INST_ANY'), ('parked_at_waitcnt', 'SQ_WAIT_ANY'), ('issue_stalled', 'SQ_WAIT_INST_ANY')) if n in c}
        d['occupancy_waves_per_simd'] = w * 4 / simd_cyc
    if 'TCC_HIT_sum' in c and 'TCC_MISS_sum' in c:
        d['l2_hit_rate'] = c['TCC_HIT_sum'] / max(c['TCC_HIT_sum'] + c['TCC_MISS_sum'], 1.0)
    if 'TCP_TOTAL_CACHE_ACCESSES_sum' in c and 'TCP_TCC_READ_REQ_sum' in c:
        d['l1_hit_rate'] = 1.0 - c['TCP_TCC_READ_REQ_sum'] / max(c['TCP_TOTAL_CACHE_ACCESSES_sum'], 1.0)
    if 'SQ_LDS_BANK_CONFLICT' in c and c.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_bank_conflict_share'] = c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']
    best = max(util.items(), key=lambda kv: kv[1]) if util else None
    return d, best


BUSY_OF = {   # resource -> (counter, busy-cycle scale, peak units per second, unit) for bench.py's live-time fraction
    'valu_issue': ('SQ_ACTIVE_INST_VALU', 4.0, N_SIMD * PEAK_CLOCK_HZ, 'SIMD-cycles/s'),
    'mfma_pipe': ('SQ_VALU_MFMA_BUSY_CYCLES', 1.0, N_SIMD * PEAK_CLOCK_HZ, 'SIMD-cycles/s'),
    'lds_array': ('SQ_LDS_IDX_ACTIVE', 1.0, N_CU * PEAK_CLOCK_HZ, 'CU-cycles/s'),
    'texture_addresser': ('TA_TA_BUSY_sum', 1.0, N_CU * PEAK_CLOCK_HZ, 'CU-cycles/s'),
    'l2_bandwidth': ('TCP_TCC_READ_REQ_sum', float(LINE), L2_PEAK_GBS * 1e9, 'B/s'),
}


def main():
    dataset = os.environ.get('P3D_PMC_DATASET', 'seg2cat')         # another BASELINE configuration's launch (tests/gpu_profile_render.py): file names get the suffix
    only = os.environ.get('P3D_PMC_GROUPS')                         # e.g. "0,1,2,3,4": a subset of the passes
    out_root = os.path.join(ROOT, 'gpurun_out', 'rpmc' + ('_exact' if os.environ.get('P3D_MLP_BF16X3', '1') == '0' else '') + ('' if dataset == 'seg2cat' else '_' + dataset))
    counters, ms, errors = {}, {}, {}
    for i, grp in enumerate(GROUPS):
        if only and str(i) not in only.split(','):
            continue
        vals, dur, err = one_pass(i, grp, out_root)
        if vals is None:
            errors[' '.join(grp)] = err
            continue
        for k, v in vals.items():
            counters[k] = v
            ms[k] = dur
    derived, best = derive(counters, ms)
    rec = {
        'kernel': 'p3d::render_forward_kernel<2, false, false, true> (bf16x3 decoder)' if os.environ.get('P3D_MLP_BF16X3', '1') != '0' else 'p3d::render_forward_kernel<2, false> (exact fp32 decoder)',
        'workload': ("bench.py's: seg2cat generator, 4 img x 128^2 rays x 64+64 samples" if dataset == 'seg2cat' else
                     f"BASELINE configs[3] per GPU: {dataset} generator, 8 img x 64^2 rays x 64+64 samples (white background, sigmoid labels)")
                    + " on the backbone's own 256^2 x 96 channels-last planes (tests/gpu_profile_render.py)",
        'source': 'rocprofv3 --kernel-trace --pmc <group>, one pass per group (tests/gpu_pmc_render.py); averages per launch of the kernel',
        'counters': counters, 'kernel_ms_in_pass': ms, 'derived': derived,
        'binding': None if best is None else {
            'resource': best[0], 'utilisation_in_pmc_pass': best[1], 'counter': BUSY_OF.get(best[0], (None,))[0],
            'busy_units_per_launch': (counters[BUSY_OF[best[0]][0]] * BUSY_OF[best[0]][1]) if best[0] in BUSY_OF else None,
            'peak_units_per_s': BUSY_OF[best[0]][2] if best[0] in BUSY_OF else None, 'unit': BUSY_OF[best[0]][3] if best[0] in BUSY_OF else None,
            'note': 'utilisation = busy cycles of the resource / (instances x GRBM_GUI_ACTIVE / 8); bench.py divides busy_units_per_launch by the LIVE launch duration and by peak_units_per_s (2.4 GHz peak clock), a lower bound on the utilisation at the clock the run held'},
        'FETCH_SIZE_KiB_per_launch': counters.get('FETCH_SIZE'), 'WRITE_SIZE_KiB_per_launch': counters.get('WRITE_SIZE'),
        'correction': 'gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported',
        'traffic_bytes_per_launch': int((2 * counters['FETCH_SIZE'] + counters['WRITE_SIZE']) * 1024) if 'FETCH_SIZE' in counters and 'WRITE_SIZE' in counters else None,
        'kernel_src_sha16': kernel_source_hash(), 'errors': errors,
    }
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    tag = ('_exact_fp32' if os.environ.get('P3D_MLP_BF16X3', '1') == '0' else '') + ('' if dataset == 'seg2cat' else '_' + dataset)
    json.dump(rec, open(os.path.join(ROOT, 'gpurun_out', f'render_pmc{tag}.json'), 'w'), indent=1)
    with open(os.path.join(ROOT, 'gpurun_out', f'render_sq_pmc{tag}.txt'), 'w') as f:
        f.write(f"# {rec['kernel']}\n# {rec['workload']}\n# kernel sources sha16 {rec['kernel_src_sha16']}\n")
        for k in sorted(counters):
            f.write(f'{k:36s} {counters[k]:18.1f}   (kernel {ms[k]:.4f} ms in that pass)\n' if ms[k] else f'{k:36s} {counters[k]:18.1f}\n')
        f.write('\n# derived\n' + json.dumps(derived, indent=1) + '\n# binding\n' + json.dumps(rec['binding'], indent=1) + '\n')
        for k, v in errors.items():
            f.write(f'# FAILED PASS [{k}]: {v[-300:]}\n')
    print(json.dumps({'binding': rec['binding'], 'derived': derived, 'traffic': rec['traffic_bytes_per_launch'], 'errors': list(errors)}))
    import shutil
    shutil.rmtree(out_root, ignore_errors=True)


if __name__ == '__main__':
    main()
