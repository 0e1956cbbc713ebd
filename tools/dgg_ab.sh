#!/bin/bash
# timing experiment (GPU box): per-graph dgrad kernel (mlp_x / mlp_m launches separately) of a training step per arm.
# arm = label[,VAR=value...]; label "base" = the built library, otherwise diffusion_model_amd/exp_<label>.so
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for arm in "$@"; do
  IFS=, read -ra parts <<< "$arm"
  label=${parts[0]}
  envs=("${parts[@]:1}")
  lib=$root/diffusion_model_amd/exp_$label.so
  out=$root/gpurun_out/dggab/$(echo "$arm" | tr ',=' '__'); mkdir -p "$out"
  (
    for e in "${envs[@]}"; do export "$e"; done
    [ -f "$lib" ] && export EGNN_LIB=$lib
    rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 "$root/bench.py" --mode train --steps 3 --warmup 1 --no-cpu-baseline > "$out/log.txt" 2>&1
  )
  f=$(find "$out" -name "*kernel_trace.csv" | head -1)
  echo "== $arm  $(grep -o '"ms_per_step": [0-9.]*' "$out/log.txt" | head -1)"; python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dgrad_graph" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
x, m = d[0::2], d[1::2]
print(f"  graph dgrad: mlp_x avg {sum(x)/len(x):.3f} ms (min {min(x):.3f}), mlp_m avg {sum(m)/len(m):.3f} ms (min {min(m):.3f}), {len(d)} launches")
PY
  find "$out" -name "*.csv" -size +4M -delete
done
