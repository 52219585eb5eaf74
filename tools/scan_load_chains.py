"""Static scan of the kernel library's ISA for DEPENDENT MEMORY ROUND TRIPS: runs of `load(s); s_waitcnt vmcnt(0)` one after another.

hipcc waits for a load that sits under a null / bounds branch INSIDE the branch (`cond ? p[i] : 0`, `if (inside) acc += p[i] * w`), keeps a read-modify-write behind the
previous store (it may alias), and does not unroll a runtime-trip loop whose body is one load — each such load is then a full memory round trip (0.5 - 2 us) with nothing in
flight beside it.  Counters only show the symptom ("waves at s_waitcnt"); this lists where the chains are.  Round 6 found and removed them in the epilogues of the patch
kernels, the split-K finish, fir4_cl_fused_kernel, upfirdn2d_cl_kernel, modulate_weights_kernel, the FC / demodulation kernels (profiles/round6_z{i,j,n,o,p,q}_*).

    python tools/scan_load_chains.py [min_chain]        (no GPU needed: hipcc -S for gfx950; prints kernel, longest chain, and the number of ROLLED loops — a backward branch
                                                         over a short body — that hold a load and a full vmcnt drain: one round trip per iteration)

A reported chain is the STATIC worst case — it may sit on a path a launch never takes (a null noise pointer skips its branch); read the kernel before acting on it."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'pix2pix3d_amd', 'csrc')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-gpu-rdc', '-DNDEBUG', '--cuda-device-only', '-S']


def scan(path, min_chain):
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, 'k.s')
        r = subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-o', asm, path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stdout[-2000:])
        name, seq = None, []
        labels, body, loops = {}, [], 0                     # rolled loops: a backward branch over a short body that holds a load and a full vmcnt drain
        for line in open(asm):
            m = re.match(r'^(_Z\w+):', line)
            if m:
                name, seq = m.group(1), []
                labels, body, loops = {}, [], 0
                continue
            if name is None:
                continue
            t = line.strip()
            ml = re.match(r'^(\.LBB\w+):', t)
            if ml:
                labels[ml.group(1)] = len(body)
            body.append(t)
            mb = re.match(r'^s_cbranch_\w+\s+(\.LBB\w+)', t)
            if mb and mb.group(1) in labels and len(body) - labels[mb.group(1)] <= 120:
                inner = body[labels[mb.group(1)]:]
                if any(x.startswith(('global_load', 'buffer_load', 'flat_load')) and 'lds' not in x.split()[0] for x in inner) and any(x.startswith('s_waitcnt') and 'vmcnt(0)' in x for x in inner):
                    loops += 1
            if t.startswith(('global_load', 'buffer_load', 'flat_load')) and 'lds' not in t.split()[0]:
                seq.append('L')
            elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
                seq.append('W')
            elif t.startswith('s_endpgm'):
                runs = re.findall(r'(?:L{1,2}W){%d,}' % min_chain, ''.join(seq))
                worst = max((r.count('W') for r in runs), default=0)
                if worst or loops:
                    out.append((worst, loops, subprocess.run(['c++filt', name], stdout=subprocess.PIPE, text=True).stdout.strip()))
                name = None
    return sorted(out, reverse=True)


if __name__ == '__main__':
    min_chain = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for src in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
        for worst, loops, kernel in scan(src, min_chain):
            print(f'{os.path.basename(src):22s} chain {worst:4d}  rolled load loops {loops:2d}  {kernel[:150]}')
