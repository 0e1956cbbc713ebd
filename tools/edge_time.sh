#!/bin/bash
# usage: tools/edge_time.sh <lib-variant>:<EGNN_DEBUG> ...   (runs on the GPU box)
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  lib=${spec%%:*}; d=${spec##*:}
  if [ "$lib" = "base" ]; then unset EGNN_LIB; else export EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_$lib.so; fi
  rm -rf /tmp/pp; EGNN_EDGE=${EDGE:-4} EGNN_DEBUG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --reps 1 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg > /dev/null 2>&1
  python3 - "$spec" <<'PY'
import csv,glob,sys
for f in glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True):
    out=[]
    for r in csv.DictReader(open(f)):
        n=r['Name'].replace('(anonymous namespace)::','')
        if 'edge_kernel' in n: out.append("%s=%.3f" % (n[n.find('edge_kernel'):n.find('(')], float(r['AverageNs'])/1e6))
    print("exp", sys.argv[1], " ".join(sorted(out)))
PY
done
