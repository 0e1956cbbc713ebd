#!/bin/bash
# second graph form (two workgroups per CU): gradient tests with it, kernel times and the training step, against the first form
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04p
EGNN_DGRAD_GRAPH2=1 timeout -k 10 420 python -m pytest tests/test_training.py -m gpu -x -q -k "full_width_64 or graph_form or uninitialised or fully_written" > gpurun_out/r04p/tests.log 2>&1; rc=$?
echo "pytest (form 2) rc=$rc"; tail -3 gpurun_out/r04p/tests.log
[ $rc -eq 0 ] || exit $rc
for f in 0 1 0 1; do
  EGNN_DGRAD_GRAPH2=$f timeout -k 10 200 python bench.py --mode train --steps 10 --warmup 3 2> gpurun_out/r04p/train.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('form2=$f train ms_per_step', round(d['ms_per_step'],2))" || exit 1
done
