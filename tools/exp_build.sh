#!/bin/bash
# Build timing-experiment variants of the library: tools/exp_build.sh <name> <extra hipcc flags...>
# -> diffusion_model_amd/exp_<name>.so  (select with EGNN_LIB=... ; results are wrong by construction)
name=$1; shift
C=diffusion_model_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -shared -DEGNN_DIAG "$@" \
  $C/egnn_forward.hip $C/edge_bf16_v3.hip $C/edge_bf16_v4.hip $C/edge_x_m16.hip $C/edge_small.hip $C/edge_bf16x3.hip $C/edge_f16c8.hip $C/edge_f16c8w.hip $C/edge_bwd_dgrad.hip $C/edge_bwd_dgrad_graph.hip $C/edge_bwd_heads.hip $C/edge_bwd_first.hip $C/gemm_tn.hip $C/gemm_rows.hip $C/sampler.hip $C/graph_stats.hip $C/aux_mlp.hip $C/node_bf16.hip $C/backward.hip $C/host_logic.cpp -o diffusion_model_amd/exp_$name.so
