#!/bin/bash
# round 4, visit a: fp16 precision -- tests, error tables (current tree, fp32 node MLP variant, the round-3 A/B tree), timing
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r04a; mkdir -p $o
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fp16 or bf16x3_matches or c2_properties" > $o/tests.log 2>&1; echo "pytest rc=$?" | tee -a $o/tests.log
tail -5 $o/tests.log
python tools/prec_errors.py --out $o/prec_errors.log > /dev/null 2> $o/prec.err; echo "prec rc=$?"
EGNN_F16_NODE=0 python tools/prec_errors.py --out $o/prec_errors_f32node.log --precisions fp16 > /dev/null 2>> $o/prec.err; echo "prec2 rc=$?"
EGNN_TREE=$GRAFT_REPO_ROOT/old_r3 EGNN_XM16=0 python tools/prec_errors.py --out $o/prec_old_32x32.log --precisions bf16 --skip-goldens > /dev/null 2>> $o/prec.err; echo "old0 rc=$?"
EGNN_TREE=$GRAFT_REPO_ROOT/old_r3 EGNN_XM16=1 python tools/prec_errors.py --out $o/prec_old_16x16.log --precisions bf16 --skip-goldens > /dev/null 2>> $o/prec.err; echo "old1 rc=$?"
for p in bf16 fp16 bf16 fp16; do
  python bench.py --precision $p --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$p', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), d['nonfinite_graphs'])" | tee -a $o/ab_fp16.log
done
tail -30 $o/prec_errors.log
