#!/bin/bash
# f16c8 edge kernels: where the time goes -- timing-only builds (wrong results by construction) against the product library,
# one box, one visit: rocprofv3 kernel-trace averages of the coordinate (X) and message (M) kernel per variant.
# usage (GPU box): bash tools/c8_ab.sh   (the exp_c8_*.so were built in the container: tools/c8_build.sh)
cd /tmp && export TMPDIR=/tmp EGNN_C8_TILE=${EGNN_C8_TILE:-32}
run() {   # name lib EGNN_DEBUG
  if [ "$2" = "base" ]; then unset EGNN_LIB; else export EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_$2.so; fi
  rm -rf /tmp/pp; EGNN_DEBUG=$3 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/bench.py --precision f16c8 --steps 5 --warmup 2 --reps 1 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs > /dev/null 2>&1
  python3 - "$1" <<'PY'
import csv,glob,sys
for f in glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True):
    out={}
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if 'edge_c8' in n: out['X' if (('<false' in n or 'ILb0' in n) and 'c8wk' not in n) else 'M']=float(r['AverageNs'])/1e6
    print("c8 %-28s X %.3f ms  M %.3f ms" % (sys.argv[1], out.get('X',0), out.get('M',0)))
PY
}
run "product" base 0
run "diag build, complete" c8_full 0
run "no correction MFMAs" c8_nocorr 0
run "no fp16 MFMAs" c8_nomain 0
run "no MFMAs at all" c8_nomfma 0
run "no activation build" c8_nobuild 0
run "no e4m3 conversions/stores" c8_nocvt8 0
run "no epilogue" c8_noepi 0
run "no weight stream" c8_full 1
run "no table rows" c8_full 2
run "no weights, no table" c8_full 3
run "product again" base 0
run "no build, no weights, no table" c8_nobuild 3
run "no build/weights/table/epilogue" c8_nobuild_noepi 3
