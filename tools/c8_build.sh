#!/bin/bash
# builds the timing-only variants of tools/c8_ab.sh (in the container; the .so files travel with the snapshot)
for v in "c8_full" "c8_nocorr -DEGNN_EXP_C8_NOCORR" "c8_nomain -DEGNN_EXP_C8_NOMAIN" "c8_nomfma -DEGNN_EXP_C8_NOCORR -DEGNN_EXP_C8_NOMAIN" \
         "c8_nobuild -DEGNN_EXP_C8_NOBUILD" "c8_nocvt8 -DEGNN_EXP_C8_NOCVT8" "c8_noepi -DEGNN_EXP_C8_NOEPI"; do
  set -- $v
  bash tools/exp_build.sh "$@" &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
ls -la diffusion_model_amd/exp_c8_*.so
