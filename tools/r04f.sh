#!/bin/bash
# round 4, visit f: factorised first-layer backward (edge_bwd_first.hip): gradient tests, then A/B of the training step
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r04f; mkdir -p $o
python -m pytest tests/test_training.py -m gpu -x -q -s > $o/tests.log 2>&1; echo "pytest rc=$?" | tee -a $o/tests.log
grep -E "gradients vs oracle|passed|failed|Error|assert" $o/tests.log | tail -15
for r in 1 2; do
for f in 0 1; do
  EGNN_BWD_FIRST=$f python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('first=$f', round(d['ms_per_step'],3), d['roofline']['backward_path'], d['final_loss'])" | tee -a $o/ab_first.log
done
done
