#!/bin/bash
# Print VGPR/AGPR/spill/LDS/occupancy per kernel of one .hip file (compile-only, no GPU needed).
f=${1:-diffusion_model_amd/csrc/egnn_forward.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|VGPRs Spill|ScratchSize|Occupancy" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(line)print line; line=$0; next}{line=line" | "$0}END{print line}'
