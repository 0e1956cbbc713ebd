#!/bin/bash
# graph-form dgrad iteration: gradient tests, kernel times (x / m) and the training step
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04k
timeout -k 10 420 python -m pytest tests/test_training.py -m gpu -x -q -k "full_width_64" > gpurun_out/r04k/tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r04k/tests.log; grep "gradients vs" gpurun_out/r04k/tests.log | tail -8
[ $rc -eq 0 ] || exit $rc
bash tools/dgg_ab.sh base "$@" || exit 1
for i in 1 2; do
  timeout -k 10 200 python bench.py --mode train --steps 10 --warmup 3 2> gpurun_out/r04k/train.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train ms_per_step', round(d['ms_per_step'],2))" || exit 1
done
