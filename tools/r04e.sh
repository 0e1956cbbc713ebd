#!/bin/bash
# round 4, visit e: 32-edge tiles for small graphs (edge_small.hip): parity on the small full-width goldens, latency A/B vs the
# 128-edge tiles (EGNN_SMALL_EDGES=0) at several thresholds
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r04e; mkdir -p $o
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "bf16_close or fp16 or deterministic or generic_edge or config0 or sampler or graph_replay or hipgraph" > $o/tests.log 2>&1; echo "pytest rc=$?" | tee -a $o/tests.log
grep -E "^bf16|^fp16|passed|failed|Error|assert" $o/tests.log | tail -20
for r in 1 2; do
for lim in 0 8192 24000; do
  EGNN_SMALL_EDGES=$lim python bench.py --steps 5 --warmup 2 --reps 1 --no-cpu-baseline --no-train-leg --no-slab-leg --no-precision-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('limit $lim', [(round(r['eager_ms_per_step'],4), round(r['graph_replay_ms_per_step'],4)) for r in d['latency']['rows']], [r['nonfinite_graphs'] for r in d['latency']['rows']])" | tee -a $o/lat_ab.log
done
done
