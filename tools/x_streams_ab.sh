#!/bin/bash
# timing experiment (GPU box): the C2 step with the coordinate / message kernels' table gathers or weight streams answered
# without leaving the CU (diagnostic build, EGNN_DEBUG bits: 1 = weights, 2 = first-layer table)
cd "$GRAFT_REPO_ROOT"
export EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_diag.so
for r in 1 2; do
  for dbg in 0 2 1 3; do
    EGNN_DEBUG=$dbg python bench.py --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EGNN_DEBUG=$dbg', 'ms_per_step', round(d['ms_per_step'],3), 'edge pass ms', round(d['roofline']['avg_launch_ms'],4), 'nonfinite', d.get('nonfinite_graphs'))"
  done
done
