#!/bin/bash
# node-backward batching iteration: the tests that cover it, then the training step twice
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04l
timeout -k 10 600 python -m pytest tests/test_training.py tests/test_gpu_round3.py -m gpu -x -q -k "gradients or gemm_rows or half_precision or saved_activation or train_step" > gpurun_out/r04l/tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r04l/tests.log; grep "gradients vs" gpurun_out/r04l/tests.log | tail -20
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
  timeout -k 10 200 python bench.py --mode train --steps 10 --warmup 3 2> gpurun_out/r04l/train.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train ms_per_step', round(d['ms_per_step'],2))" || exit 1
done
