"""Memory-side traffic of the fused ray-marcher at the bench workload, as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes:
two separate `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE, WRITE_SIZE; no other trace domain) over
tests/gpu_profile_render.py, FETCH_SIZE doubled (gfx950 tallies its 128-byte requests at 64 B), per launch.  Writes
gpurun_out/render_pmc.json (copy it to profiles/): bench.py reports `roofline.traffic` from it only while the recorded hash of the
kernel sources still matches the tree being benchmarked.

    python tests/gpu_pmc_traffic.py            (on the GPU box)
"""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ['pix2pix3d_amd/csrc/render.hip', 'pix2pix3d_amd/csrc/render_device.h']


def kernel_source_hash(root=ROOT):
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(root, rel), 'rb').read())
    return h.hexdigest()[:16]


def one_pass(counter, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, REPS='1', TMPDIR='/tmp')
    cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', out_dir, '-o', 'r', '--',
           sys.executable, os.path.join(ROOT, 'tests', 'gpu_profile_render.py')]
    r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    files = glob.glob(os.path.join(out_dir, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise RuntimeError(f'{counter}: no counter file\n{r.stdout[-2000:]}')
    vals = []
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if 'render_forward_kernel' in row.get('Kernel_Name', '') and row.get('Counter_Name') == counter:
                vals.append(float(row['Counter_Value']))
    if not vals:
        raise RuntimeError(f'{counter}: kernel not found in {files[0]}')
    return sum(vals) / len(vals), len(vals), files[0]


def main():
    out = os.path.join(ROOT, 'gpurun_out', 'rpmc_traffic')
    fetch, nf, ff = one_pass('FETCH_SIZE', os.path.join(out, 'fetch'))
    write, nw, wf = one_pass('WRITE_SIZE', os.path.join(out, 'write'))
    rec = {
        'kernel': 'p3d::render_forward_kernel<2, false>',
        'workload': '4 img x 128^2 rays x 64+64 samples, 256^2x96 channels-last planes (tests/gpu_profile_render.py, random planes)',
        'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tests/gpu_pmc_traffic.py)',
        'FETCH_SIZE_KiB_per_launch': fetch, 'WRITE_SIZE_KiB_per_launch': write, 'launches_averaged': [nf, nw],
        'correction': 'gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported',
        'traffic_bytes_per_launch': int((2 * fetch + write) * 1024),
        'kernel_src_sha16': kernel_source_hash(),
        'note': 'memory-side (fabric) requests of the L2s, Infinity-Cache hits included',
    }
    path = os.path.join(ROOT, 'gpurun_out', 'render_pmc.json')
    json.dump(rec, open(path, 'w'), indent=2)
    for src, name in ((ff, 'render_pmc_FETCH_SIZE.csv'), (wf, 'render_pmc_WRITE_SIZE.csv')):
        rows = [l for l in open(src) if 'render_forward_kernel' in l or l.startswith('"Correlation') or l.startswith('Correlation')]
        open(os.path.join(ROOT, 'gpurun_out', name), 'w').writelines(rows)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
