#!/bin/bash
# per-kernel durations of one reverse step at B = 1 x 64 atoms (the reference's per-call workload); run through gpurun
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -- python3 $GRAFT_REPO_ROOT/bench.py --batch ${1:-1} --steps 40 --warmup 5 --reps 1 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/lp/**/*kernel_stats.csv', recursive=True):
    tot=0
    for r in csv.DictReader(open(f)):
        n=r['Name'].replace('(anonymous namespace)::','')
        print("%-70s calls=%5s avg_us=%8.1f" % (n[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
