#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: kernel trace + PMC passes of bench.py.
# Usage: tools/profile.sh <tag> [bench args...]
set -o pipefail
tag=${1:-r01}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--steps 5 --warmup 2 --reps 1 --no-cpu-baseline --no-train-leg --no-slab-leg --no-latency-leg --no-precision-legs $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" $args > "$out/trace.log" 2>&1
echo "trace rc=$?" >> "$out/trace.log"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d "$out/pmc_sq" -- python3 "$root/bench.py" $args > "$out/pmc_sq.log" 2>&1
echo "pmc_sq rc=$?" >> "$out/pmc_sq.log"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$out/pmc_sq2" -- python3 "$root/bench.py" $args > "$out/pmc_sq2.log" 2>&1
echo "pmc_sq2 rc=$?" >> "$out/pmc_sq2.log"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_fetch" -- python3 "$root/bench.py" $args > "$out/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?" >> "$out/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_write" -- python3 "$root/bench.py" $args > "$out/pmc_write.log" 2>&1
echo "pmc_write rc=$?" >> "$out/pmc_write.log"
python3 "$root/tools/summarize_prof.py" "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
python3 "$root/tools/make_traffic_json.py" "$out" "${GIT_HEAD:-unknown}" "$out/traffic.json" > /dev/null 2>&1
# keep the merged-back directory small
find "$out" -name "*.csv" -size +8M -delete
