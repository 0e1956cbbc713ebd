#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: kernel trace of bench.py --mode train, per-step table.
# Usage: tools/profile_train.sh <tag> [bench args...]
set -o pipefail
tag=${1:-train}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
steps=3; warm=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --mode train --steps $steps --warmup $warm --no-cpu-baseline "$@" > "$out/trace.log" 2>&1
echo "trace rc=$?" >> "$out/trace.log"
tail -2 "$out/trace.log"
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
python3 - "$f" $((steps + warm)) > "$out/summary.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per training step (averaged over {n} steps incl. warmup): {tot / 1e6 / n:.2f} ms")
for r in rows[:30]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / n:9.3f} ms/step {int(r['Calls']) / n:8.1f} calls/step "
          f"{float(r['Percentage']):5.1f}%  {r['Name'][:120]}")
PY
cat "$out/summary.txt"
find "$out" -name "*.csv" -size +8M -delete
