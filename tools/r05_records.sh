#!/bin/bash
# records of a round-5 state (usage on the GPU box: bash tools/r05_records.sh <tag>): profiles, precision log, bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r05S}
mkdir -p gpurun_out/$tag
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/profile.sh $tag > gpurun_out/$tag/profile.log 2>&1; echo "profile rc=$?"
cp gpurun_out/prof_$tag/traffic.json profiles/traffic.json 2>/dev/null
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/profile_train_pmc.sh ${tag}_train > gpurun_out/$tag/profile_train.log 2>&1; echo "profile_train rc=$?"
cp gpurun_out/prof_${tag}_train/traffic_train.json profiles/traffic_train.json 2>/dev/null
python tools/prec_errors.py --out gpurun_out/$tag/prec_errors.log > gpurun_out/$tag/prec.out 2>&1; cp gpurun_out/$tag/prec_errors.log profiles/${tag}_prec_errors.log; grep golden_max gpurun_out/$tag/prec.out
python bench.py > gpurun_out/$tag/bench_c2.json 2> gpurun_out/$tag/bench_c2.err; echo "c2 rc=$?"
python bench.py --atoms 512 --batch 32 --steps 5 --warmup 2 --reps 3 --no-cpu-baseline > gpurun_out/$tag/bench_c3.json 2> gpurun_out/$tag/bench_c3.err; echo "c3 rc=$?"
python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/$tag/bench_train.json 2> gpurun_out/$tag/bench_train.err; echo "train rc=$?"
BENCH_DEVICE=0 BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --reps 2 --train-steps 3 > gpurun_out/$tag/bench_2rank_gloo.json 2> gpurun_out/$tag/bench_2rank_gloo.err; echo "2rank rc=$?"
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d=json.load(open(f"gpurun_out/{tag}/bench_c2.json"))
print("c2", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("golden_max_rel_err"))
t=d["tolerance_grade"]; print("tg", t["precision"], t["ms_per_step"], t["golden_max_rel_err"], t["meets_tolerance"], t["candidates"])
print("train", d["ddp_train"]["ms_per_step"], d["ddp_train"]["roofline"]["traffic"], "tg-train", d["ddp_train_tolerance_grade"]["ms_per_step"])
print("c3", d["c3"]["ms_per_step"], d["c3"]["metric"])
r=json.loads([l for l in open(f"gpurun_out/{tag}/bench_2rank_gloo.json") if l.startswith("{")][-1])
print("2rank", r["ms_per_step"], r.get("ddp_efficiency"), r["ddp_train"].get("ranks_share_device"), r["ddp_train"].get("ddp_efficiency_basis"), r["ddp_train"].get("efficiency_vs_no_collective"))
PY
