#!/bin/bash
# interleaved A/B of f16c8 library variants (diffusion_model_amd/exp_<name>.so, built in the container) against the product library:
# edge pass ms (HIP events) and ms per C2 step.   usage (GPU box): bash tools/c8_ab2.sh <rounds> <name> [<name> ...]
cd "$GRAFT_REPO_ROOT"
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for arm in base "$@"; do
    if [ "$arm" = "base" ]; then unset EGNN_LIB; else export EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_$arm.so; fi
    python bench.py --precision f16c8 --steps 10 --warmup 3 --reps 2 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-18s ms/step %.3f  edge pass %.4f ms  nonfinite %d' % ('$arm', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['nonfinite_graphs']))"
  done
done
