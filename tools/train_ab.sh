#!/bin/bash
# interleaved A/B of library variants on the bf16 TRAINING step (bench.py --mode train): usage (GPU box): bash tools/train_ab.sh <rounds> <exp name> [...]
cd "$GRAFT_REPO_ROOT"
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for arm in "$@"; do
    if [ "$arm" = "base" ]; then unset EGNN_LIB; else export EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_$arm.so; fi
    python bench.py --mode train --steps 10 --warmup 3 --reps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-14s train ms/step %.3f' % ('$arm', d['ms_per_step']))"
  done
done
