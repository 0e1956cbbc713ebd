#!/bin/bash
# round 4, visit c: full GPU suite on the hygiene build (hidden visibility, host_logic.cpp, diag.h, buffer-store SAVE) + the new bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r04c; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $o/gpu_tests.log
tail -4 $o/gpu_tests.log
python bench.py > $o/bench_c2.json 2> $o/bench_c2.err; echo "c2 rc=$?"
python bench.py --mode train --steps 10 --warmup 3 > $o/bench_train.json 2> $o/bench_train.err; echo "train rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c/bench_c2.json').read().strip().splitlines()[-1])
print('headline', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
for k in ('tolerance_grade','fp16','c3'):
    r=d[k]; print(k, r['ms_per_step'], r['value'], r['roofline']['frac'], r.get('golden_max_rel_err'))
print('tolerance_grade.c3', d['tolerance_grade']['c3']['ms_per_step'])
print('train', d['ddp_train']['ms_per_step'])
t=json.loads(open('gpurun_out/r04c/bench_train.json').read().strip().splitlines()[-1]); print('train leg', t['ms_per_step'], t['roofline']['backward_path'])
PY
