#!/bin/bash
# training forward (activation-keeping coordinate kernel edge_x_m16_kernel<true>): which store costs what -- timing builds
# (wrong gradients by construction) against the diag baseline, rocprofv3 kernel averages of bench.py --mode train
cd /tmp && export TMPDIR=/tmp
for v in ${TF_ARMS:-tf_base tf_nos1 tf_nostage tf_none}; do
  rm -rf /tmp/pp; EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 3 --warmup 1 --reps 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - "$v" <<'PY'
import csv,glob,sys
for f in glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if 'edge_x_m16' in n or 'bf16_v4' in n: print("%-12s %-70s avg %.4f ms  calls %s" % (sys.argv[1], n[:70], float(r['AverageNs'])/1e6, r['Calls']))
PY
done
