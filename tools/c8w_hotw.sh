#!/bin/bash
# f16c8 32x32 coordinate kernel: what does the weight stream cost -- misses of the L2 or the path from the L2?  EGNN_DEBUG bit 2 =
# every chunk reads chunk 0's fragments (always L2 hits), bit 0 = no weight loads at all; diag build exp_c8_full.so
cd /tmp && export TMPDIR=/tmp EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_c8_full.so
for dbg in 0 4 1 0 4 1; do
  rm -rf /tmp/pp; EGNN_DEBUG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/bench.py --precision f16c8 --steps 5 --warmup 2 --reps 1 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs > /dev/null 2>&1
  python3 - "$dbg" <<'PY'
import csv,glob,sys
for f in glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True):
    out={}
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if 'edge_c8' in n: out['X' if (('<false' in n or 'ILb0' in n) and 'c8wk' not in n) else 'M']=float(r['AverageNs'])/1e6
    print("EGNN_DEBUG=%s: X %.3f ms  M %.3f ms" % (sys.argv[1], out.get('X',0), out.get('M',0)))
PY
done
