#!/bin/bash
# graph-form dgrad: gradient tests, then the training step with / without it on one box
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04j
timeout -k 10 420 python -m pytest tests/test_training.py -m gpu -x -q -k "full_width_64 or half_precision or saved_activation" > gpurun_out/r04j/tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -15 gpurun_out/r04j/tests.log
[ $rc -eq 0 ] || exit $rc
for g in 1 0 1 0; do
  EGNN_BWD_GRAPH=$g timeout -k 10 200 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/r04j/train_g$g.json 2> gpurun_out/r04j/train_g$g.err || exit 1
  echo "graph=$g"; tail -c 300 gpurun_out/r04j/train_g$g.json; echo
done
