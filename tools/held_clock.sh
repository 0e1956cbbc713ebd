#!/bin/bash
# In-kernel (held) clock of the coordinate kernel's K loop for both MFMA shapes: d(s_memtime) / d(s_memrealtime) x 100 MHz
# from the stamp build (diagnostic library: no stamp executes in the product), after 200 back-to-back steps on random data.
# usage (GPU box): bash tools/held_clock.sh
cd "$GRAFT_REPO_ROOT" || exit 1
[ -f diffusion_model_amd/exp_stamp.so ] || bash tools/exp_build.sh stamp -DEGNN_EXP_STAMP || exit 1   # (built in the container: travels with the snapshot)
for arm in 1; do   # (the 32x32x16 arm, EGNN_XM16=0, existed up to commit "16x16x32 coordinate kernel: training-forward (SAVE) mode": profiles/r03e_held_clock.txt)
  echo "== coordinate kernel edge_x_m16_kernel =="
  EGNN_LIB=$GRAFT_REPO_ROOT/diffusion_model_amd/exp_stamp.so python3 tools/stamps.py 2>/dev/null | awk '/^kernel M/{exit} {print}' | grep -E "kernel X|tile anatomy|in-kernel clock"
done
