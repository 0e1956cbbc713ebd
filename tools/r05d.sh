#!/bin/bash
# r05d: precision f16c8 after a kernel change -- golden errors, then the C2 bench line (edge pass ms, step ms)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r05d}
mkdir -p gpurun_out/$tag
timeout -k 10 400 python tools/prec_errors.py --out gpurun_out/$tag/prec_errors.log --precisions f16c8 --skip-c2 > gpurun_out/$tag/prec.out 2>&1; echo "prec rc=$?"
grep "full_\|golden_max" gpurun_out/$tag/prec.out
timeout -k 10 300 python bench.py --precision f16c8 --steps 10 --warmup 3 --reps 3 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs > gpurun_out/$tag/bench_f16c8.json 2> gpurun_out/$tag/bench_f16c8.err; echo "f16c8 rc=$?"
python -c "import json;d=json.load(open('gpurun_out/$tag/bench_f16c8.json'));print('ms/step', d['ms_per_step'], 'edge pass', d['roofline']['avg_launch_ms'], 'nonfinite', d['nonfinite_graphs'], 'node', d['node_kernels_ms_per_layer'])"
