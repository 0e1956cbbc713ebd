#!/bin/bash
# interleaved A/B of library variants on the HEADLINE (bf16) line: usage (GPU box): bash tools/ab_bf16.sh <rounds> <exp name> [...]
cd "$GRAFT_REPO_ROOT"
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for arm in base "$@"; do
    if [ "$arm" = "base" ]; then unset EGNN_LIB; else export EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_$arm.so; fi
    python bench.py --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-14s ms/step %.3f  edge pass %.4f ms  frac %.4f' % ('$arm', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
  done
done
