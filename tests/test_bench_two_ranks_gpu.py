"""bench.py's own six-phase training iteration with two ranks (one GPU, backend gloo on device tensors: RCCL refuses two ranks on one device): after
the iteration the replicas hold bit-identical parameters although they saw different data — the flat gradient exchange and the optimizer step of
`bench.py --gpus N --train-step` do what training_loop.py:528-543 does.  Worker: tests/bench_two_ranks_worker.py."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_training_iteration_keeps_two_replicas_in_lock_step(hip_lib):
    port = 29900 + os.getpid() % 90
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'bench_two_ranks_worker.py')], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    if not all(p.returncode == 0 for p in procs):
        for r, o in enumerate(outs):
            print(f'---- rank {r} (rc {procs[r].returncode}) ----\n' + '\n'.join(l for l in o.splitlines() if 'socket.cpp' not in l)[-4000:])
    assert all(p.returncode == 0 for p in procs), [p.returncode for p in procs]
    assert 'BENCH_TWO_RANKS_OK' in outs[0], outs[0][-2000:]
    print(outs[0].strip().splitlines()[-1])
