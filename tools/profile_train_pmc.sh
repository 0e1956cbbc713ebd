#!/bin/bash
# Run on the GPU box: PMC passes of the training step (bench.py --mode train), per-kernel means for the backward kernels.
# Usage: tools/profile_train_pmc.sh <tag>
set -o pipefail
tag=${1:-trainpmc}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--mode train --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" $args > "$out/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d "$out/pmc_sq" -- python3 "$root/bench.py" $args > "$out/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$out/pmc_sq2" -- python3 "$root/bench.py" $args > "$out/pmc_sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_fetch" -- python3 "$root/bench.py" $args > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_write" -- python3 "$root/bench.py" $args > "$out/pmc_write.log" 2>&1
python3 - "$out" > "$out/summary.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("egnn::", "")
    return n.split("(")[0][-70:]
keep = ("dgrad", "heads", "edge_x", "edge_kernel", "scatter", "gather", "Cijk")
print("== kernel trace ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print(f"{short(r['Name']):70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
print("== PMC (mean per dispatch, by kernel and grid size) ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d): continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not any(x in k for x in keep): continue
            acc[(k, r.get("Grid_Size", r.get("Grid_Size_X", "")))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), cs in sorted(acc.items()):
        print(f"[{os.path.basename(d)}] {k} grid={g}")
        for c, v in sorted(cs.items()):
            print(f"    {c:30s} mean={sum(v)/len(v):.4g} n={len(v)}")
PY
cat "$out/summary.txt"
python3 "$root/tools/make_train_traffic_json.py" "$out" "${GIT_HEAD:-unknown}" 3 "$out/traffic_train.json" > /dev/null 2>&1
find "$out" -name "*.csv" -size +4M -delete
