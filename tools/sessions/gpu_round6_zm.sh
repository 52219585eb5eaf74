#!/bin/bash
# round-6 session zm: the default line of the final tree once more (the pool's boxes differ by +-4 %; session zk drew a slow one)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; T=round6_zm
( timeout 900 python bench.py 2>gpurun_out/${T}_bench.err | tail -1 ) > gpurun_out/${T}_bench_line_default.json
python -c "
import json; d = json.load(open('gpurun_out/${T}_bench_line_default.json')); e = d['exact_fp32']; t = d['train_step']
print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['ms_per_launch'], d['roofline']['frac'], 'exact', e['value'], e['backbone_as_bf16x6']['value'], 'train', t['ms_per_iteration'], t['lazy_schedule']['ms_per_iteration'], {k[:10]: v['value'] for k, v in d['configs'].items()})"
echo finished
