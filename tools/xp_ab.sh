#!/bin/bash
# persistent coordinate kernel (EGNN_X_PERSIST, edge_x_m16.hip) on / off in one library: golden errors, then interleaved headline runs
# usage (GPU box): bash tools/xp_ab.sh <rounds>
cd "$GRAFT_REPO_ROOT"
for v in 0 1; do
  echo "== EGNN_X_PERSIST=$v"; EGNN_X_PERSIST=$v timeout -k 10 300 python3 tools/prec_errors.py --precisions bf16,fp16 --skip-c2 2>&1 | tail -3 || exit 1
done
for r in $(seq 1 $1); do
  for v in 0 1; do
    EGNN_X_PERSIST=$v python bench.py --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('persist %s  ms/step %.3f  edge pass %.4f ms  frac %.4f  nonfinite %d' % ('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['nonfinite_graphs']))"
  done
done
