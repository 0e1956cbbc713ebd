#!/bin/bash
# round 4, visit b: fp16 with the split-operand node MLP (also under bf16x3): tests, error tables, timing
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r04b; mkdir -p $o
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fp16 or bf16x3 or c2_properties" > $o/tests.log 2>&1; echo "pytest rc=$?" | tee -a $o/tests.log
tail -5 $o/tests.log
python tools/prec_errors.py --out $o/prec_errors.log > /dev/null 2> $o/prec.err; echo "prec rc=$?"
for p in bf16 fp16 bf16x3 bf16 fp16 bf16x3; do
  python bench.py --precision $p --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$p', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), round(d['node_kernels_ms_per_layer'],4), d['nonfinite_graphs'])" | tee -a $o/ab_fp16.log
done
grep -E "golden_max|^C2|full_" $o/prec_errors.log
