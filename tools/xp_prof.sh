#!/bin/bash
# kernel-trace averages of the bf16 edge kernels with the persistent coordinate kernel off / on (one library, one box)
cd /tmp && export TMPDIR=/tmp
for v in 0 1 0 1; do
  rm -rf /tmp/pp; EGNN_X_PERSIST=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --reps 1 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs > /dev/null 2>&1
  python3 - "$v" <<'PY'
import csv,glob,sys
for f in glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if 'edge_x_m16' in n or 'bf16_v4' in n: print("persist %s: %-60s avg %.4f ms  calls %s" % (sys.argv[1], n[:60], float(r['AverageNs'])/1e6, r['Calls']))
PY
done
