#!/bin/bash
# precision f16c8: 32x32 matrix tiles (edge_f16c8w.hip, EGNN_C8_TILE=32) against 16x16 (edge_f16c8.hip) in the SAME process
# image and box: parity of both against the goldens, per-kernel times (rocprofv3), then interleaved ms per C2 step.
# usage (GPU box): bash tools/c8w_ab.sh <rounds>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in 16 32; do
  echo "== EGNN_C8_TILE=$t: golden errors"
  EGNN_C8_TILE=$t timeout -k 10 300 python3 $R/tools/prec_errors.py --precisions f16c8 --skip-c2 2>&1 | tail -4 || exit 1
done
for t in 16 32 32k1 32k0; do   # 32k<n>: EGNN_C8_KSPLIT=<n> (message kernel variants)
  case $t in *k*) export EGNN_C8_KSPLIT=${t#*k};; *) unset EGNN_C8_KSPLIT;; esac
  rm -rf /tmp/pp; EGNN_C8_TILE=${t%k*} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/bench.py --precision f16c8 --steps 5 --warmup 2 --reps 1 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs > /dev/null 2>&1
  python3 - "$t" <<'PY'
import csv,glob,sys
for f in glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True):
    out={}
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if 'edge_c8' in n: out['X' if (('<false' in n or 'ILb0' in n) and 'c8wk' not in n) else 'M']=float(r['AverageNs'])/1e6
    print("tile %s: X %.3f ms  M %.3f ms" % (sys.argv[1], out.get('X',0), out.get('M',0)))
PY
done
unset EGNN_C8_KSPLIT
cd $R
for r in $(seq 1 $1); do
  for t in 16 32; do
    EGNN_C8_TILE=$t python bench.py --precision f16c8 --steps 10 --warmup 3 --reps 2 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile %-4s ms/step %.3f  edge pass %.4f ms  nonfinite %d' % ('$t', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['nonfinite_graphs']))"
  done
done
