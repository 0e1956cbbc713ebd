#!/bin/bash
# round 4, visit d: the new / changed tests (gradients vs oracle per precision, bf16x3 / fp16 in the chain, statistics and
# partition tests, the full-width statistical test)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r04d; mkdir -p $o
python -m pytest tests/test_gpu_fullwidth.py -m gpu -x -q -s > $o/fullwidth.log 2>&1; echo "fullwidth rc=$?" | tee -a $o/fullwidth.log
grep -E "chain|training|vs fp32|seed 7|passed|failed|Error|assert" $o/fullwidth.log | tail -20
python -m pytest tests/test_training.py tests/test_gpu_sample_stats.py tests/test_partition.py -m gpu -x -q -s -k "gradients or half_precision or full_chain or statistics or emulated" > $o/tests.log 2>&1; echo "tests rc=$?" | tee -a $o/tests.log
grep -E "gradients vs oracle|full chain|passed|failed|Error" $o/tests.log | tail -30
