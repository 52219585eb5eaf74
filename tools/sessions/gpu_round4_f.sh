#!/bin/bash
# round 4, session f: the whole -m gpu suite on the closing tree + smoke()
tag=${1:-round4_f}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/${tag}_gputest.log 2>&1; tail -n 6 gpurun_out/${tag}_gputest.log | cut -c1-600
grep -E "^E  " gpurun_out/${tag}_gputest.log | head -10 | cut -c1-3000
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; tail -n 1 gpurun_out/${tag}_smoke.log | cut -c1-400
