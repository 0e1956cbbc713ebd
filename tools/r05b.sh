#!/bin/bash
# r05b: precision f16c8 first light -- golden errors against bf16x3, then the C2 bench lines of both
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05b
timeout -k 10 400 python tools/prec_errors.py --out gpurun_out/r05b/prec_errors.log --precisions f16c8,bf16x3 --skip-c2 > gpurun_out/r05b/prec.out 2>&1; echo "prec rc=$?"
tail -14 gpurun_out/r05b/prec.out
timeout -k 10 300 python bench.py --precision f16c8 --steps 10 --warmup 3 --reps 3 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs > gpurun_out/r05b/bench_f16c8.json 2> gpurun_out/r05b/bench_f16c8.err; echo "f16c8 rc=$?"
python -c "import json;d=json.load(open('gpurun_out/r05b/bench_f16c8.json'));print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['nonfinite_graphs'])"
timeout -k 10 300 python bench.py --precision bf16x3 --steps 10 --warmup 3 --reps 3 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs > gpurun_out/r05b/bench_bf16x3.json 2> gpurun_out/r05b/bench_bf16x3.err; echo "bf16x3 rc=$?"
python -c "import json;d=json.load(open('gpurun_out/r05b/bench_bf16x3.json'));print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['nonfinite_graphs'])"
