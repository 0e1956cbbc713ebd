#!/bin/bash
# round 4, visit h: latency work -- node_post hidden-split form with a one-round gather, 64-row small tiles for one 64-atom graph
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/r04h; mkdir -p $o
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -k "bf16_close or fp16 or deterministic or generic_edge or config0 or sampler or graph or hipgraph or x_only or bf16x3_matches" > $o/tests.log 2>&1; echo "pytest rc=$?" | tee -a $o/tests.log
tail -3 $o/tests.log
for r in 1 2; do
for cfg in "0 0" "2048 0" "2048 6144" "2048 24000"; do
  set -- $cfg
  EGNN_SMALL_EDGES=$1 EGNN_SMALL64_EDGES=$2 python bench.py --steps 5 --warmup 2 --reps 1 --no-cpu-baseline --no-train-leg --no-slab-leg --no-precision-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('small32<=$1 small64<=$2', [(round(r['eager_ms_per_step'],4), round(r['graph_replay_ms_per_step'],4)) for r in d['latency']['rows']], [r['nonfinite_graphs'] for r in d['latency']['rows']])" | tee -a $o/lat_ab.log
done
done
bash tools/latency_prof.sh 1 2>&1 | grep -E "edge_|node_|sampler_step" | head -8 | tee -a $o/lat_prof.log
