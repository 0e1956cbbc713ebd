#!/bin/bash
# interleaved A/B of two libraries with bench.py (each arm its own process): usage ab_lib.sh <exp name> <rounds>
cd "$GRAFT_REPO_ROOT"
for r in $(seq 1 ${2:-3}); do
  for arm in base $1; do
    if [ "$arm" = "base" ]; then unset EGNN_LIB; else export EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_$arm.so; fi
    python bench.py ${BENCH_ARGS:---steps 20 --warmup 5 --reps 3} --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
  done
done
