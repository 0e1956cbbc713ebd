#!/bin/bash
# timing experiment (GPU box): dgrad kernel averages of a training step per arm.
# arm = label[,VAR=value...]; label "base" = the built library, otherwise diffusion_model_amd/exp_<label>.so if it exists
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for arm in "$@"; do
  IFS=, read -ra parts <<< "$arm"
  label=${parts[0]}
  envs=("${parts[@]:1}")
  lib=$root/diffusion_model_amd/exp_$label.so
  out=$root/gpurun_out/dgab/$(echo "$arm" | tr ',=' '__'); mkdir -p "$out"
  (
    for e in "${envs[@]}"; do export "$e"; done
    [ -f "$lib" ] && export EGNN_LIB=$lib
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/bench.py" --mode train --steps 3 --warmup 1 --no-cpu-baseline > "$out/log.txt" 2>&1
  )
  f=$(find "$out" -name "*kernel_stats.csv" | head -1)
  echo "== $arm  $(grep -o '"ms_per_step": [0-9.]*' "$out/log.txt" | head -1)"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dgrad" in r["Name"] or "x_m16" in r["Name"] or "bf16_v4" in r["Name"]: print(f"  {r['Name'][40:80]:40s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:.3f} ms")
PY
  find "$out" -name "*.csv" -size +4M -delete
done
