#!/bin/bash
# the whole GPU suite with poisoned uninitialised memory (tests/conftest.py, EGNN_TEST_POISON): run on the GPU box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/poison
EGNN_TEST_POISON=1 timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/poison/tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/poison/tests.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/poison/tests.log | head -40
