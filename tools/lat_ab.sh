#!/bin/bash
# interleaved A/B of two libraries on the small-batch latency rows of bench.py: usage lat_ab.sh <exp name> <rounds>
cd "$GRAFT_REPO_ROOT"
for r in $(seq 1 ${2:-3}); do
  for arm in base $1; do
    if [ "$arm" = "base" ]; then unset EGNN_LIB; else export EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_$arm.so; fi
    python bench.py --steps 5 --warmup 2 --reps 1 --no-cpu-baseline --no-train-leg --no-slab-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', [round(r['graph_replay_ms_per_step'],4) for r in d['latency']['rows']], round(d['node_kernels_ms_per_layer'],4))"
  done
done
