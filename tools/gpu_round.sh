#!/bin/bash
# one GPU-box visit: tests, smoke, the bench lines of every single-GPU BASELINE config, the 2-rank rehearsal; optional profile pass
# usage: tools/gpu_round.sh <tag> [profile]
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04z}
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -x -q > gpurun_out/$tag/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/$tag/gpu_tests.log
tail -3 gpurun_out/$tag/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/$tag/smoke.log
if [ "$2" = "profile" ]; then
  GIT_HEAD=${GIT_HEAD:-unknown} bash tools/profile.sh $tag > gpurun_out/$tag/profile.log 2>&1; echo "profile rc=$?"
  cp gpurun_out/prof_$tag/traffic.json profiles/traffic.json 2>/dev/null   # the bench lines below quote it (fingerprint of the edge-kernel sources)
  GIT_HEAD=${GIT_HEAD:-unknown} bash tools/profile_train_pmc.sh ${tag}_train > gpurun_out/$tag/profile_train.log 2>&1; echo "profile_train rc=$?"
  cp gpurun_out/prof_${tag}_train/traffic_train.json profiles/traffic_train.json 2>/dev/null   # (fingerprint of csrc/ + autograd.py + gemm.py)
fi
# measured errors of every precision on the goldens + C2 batch, with this tree's forward-source fingerprint: the bench lines below quote it
python tools/prec_errors.py --out gpurun_out/$tag/prec_errors.log > gpurun_out/$tag/prec.out 2>&1; cp gpurun_out/$tag/prec_errors.log profiles/${tag}_prec_errors.log; grep golden_max gpurun_out/$tag/prec.out
python bench.py > gpurun_out/$tag/bench_c2.json 2> gpurun_out/$tag/bench_c2.err; echo "c2 rc=$?"
python bench.py --atoms 512 --batch 32 --steps 5 --warmup 2 --reps 3 --no-cpu-baseline > gpurun_out/$tag/bench_c3.json 2> gpurun_out/$tag/bench_c3.err; echo "c3 rc=$?"
python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/$tag/bench_train.json 2> gpurun_out/$tag/bench_train.err; echo "train rc=$?"
BENCH_DEVICE=0 BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --reps 2 --train-steps 3 > gpurun_out/$tag/bench_2rank_gloo.json 2> gpurun_out/$tag/bench_2rank_gloo.err; echo "2rank rc=$?"
tail -c 600 gpurun_out/$tag/bench_c2.json; echo
tail -c 400 gpurun_out/$tag/bench_c3.json; echo
tail -c 400 gpurun_out/$tag/bench_train.json; echo
tail -c 400 gpurun_out/$tag/bench_2rank_gloo.json; tail -5 gpurun_out/$tag/bench_2rank_gloo.err
