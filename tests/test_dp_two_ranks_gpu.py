"""Two processes through the HIP path on one GPU (world size 2, backend gloo on device tensors — RCCL refuses two ranks on one device): keeps
``bench.py --gpus N [--train-step]`` honest until an 8-GPU node yields the scaling curve.  The worker is tests/dp_two_ranks_worker.py."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_share_one_gpu(hip_lib):
    port = 29700 + os.getpid() % 200
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp_two_ranks_worker.py')], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
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
    assert 'TWO_RANKS_OK' in outs[0], outs[0][-2500:]
    print(outs[0].strip().splitlines()[-1])
