#!/bin/bash
# f16c8 coordinate kernel: the matrix phases alone (no build, no weight stream, no table rows) -- how far from the MFMA pipe time?
cd /tmp && export TMPDIR=/tmp
run() {   # name lib EGNN_DEBUG
  if [ "$2" = "base" ]; then unset EGNN_LIB; else export EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_$2.so; fi
  rm -rf /tmp/pp; EGNN_DEBUG=$3 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/bench.py --precision f16c8 --steps 5 --warmup 2 --reps 1 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs > /dev/null 2>&1
  python3 - "$1" <<'PY'
import csv,glob,sys
for f in glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True):
    out={}
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if 'edge_c8_kernel' in n: out['X' if ('<false' in n or 'ILb0' in n) else 'M']=float(r['AverageNs'])/1e6
    print("c8 %-44s X %.3f ms  M %.3f ms" % (sys.argv[1], out.get('X',0), out.get('M',0)))
PY
}
run "product" base 0
run "no build" c8_nobuild 0
run "no build, no weights, no table" c8_nobuild 3
run "no build, no weights/table, no epilogue" c8_nobuild_noepi 3
run "... and rings 4 / 3" c8_nobuild_noepi_r43 3
run "complete kernel, rings 4 / 3" c8_r43 0
