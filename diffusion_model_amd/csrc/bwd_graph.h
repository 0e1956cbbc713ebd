// Per-graph fused dgrad of the training backward (edge_bwd_dgrad_graph.hip): declarations shared by its launcher, the layer
// code that fills its arguments from the context and the C entry point.
#pragma once
#include "common.h"

namespace egnn {
int init_edge_dgrad_graph_attributes();
int launch_edge_dgrad_graph(int N, int B, const int* graph_ptr, const int* row_ptr, const int* dst, const int* src, int e_first,
                            int n_edges, const float* x, const void* table, int TC, int offP, int offQ, const float* wd,
                            const void* g_a2, int Kd, const void* w2t, int KP, void* Gd, void* Gs, int ldg, float* cd,
                            float* gd2_part, hipStream_t st);
// both edge MLPs of `layer` over the chunk [e_first, e_first + n_edges) (whole graphs of <= 64 nodes)
int backward_dgrad_graph(egnn_ctx* c, hipStream_t st, int layer, const float* x, int e_first, int n_edges, const void* g_a2x,
                         const void* g_a2m, void* G, float* cd_x, float* cd_m, float* gd2_part);
}  // namespace egnn
