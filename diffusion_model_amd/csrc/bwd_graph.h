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
// second form (edge_bwd_dgrad_graph2.hip): two workgroups per CU on 128-column slices; gd2_part = [KP / 128][n_edges]
int init_edge_dgrad_graph2_attributes();
int launch_edge_dgrad_graph2(int N, int B, const int* graph_ptr, const int* row_ptr, const int* dst, const int* src, int e_first,
                             int n_edges, const float* x, const void* table, int TC, int offP, int offQ, const float* wd,
                             const void* g_a2, int Kd, const void* w2t, int KP, void* Gd, void* Gs, int ldg, float* cd,
                             float* gd2_part, hipStream_t st);
// columns per share of dL/d(d2_e) of the form that runs (256: first form; 128: second form, EGNN_DGRAD_GRAPH2=1)
int dgrad_graph_share_columns();
// both edge MLPs of `layer` over the chunk [e_first, e_first + n_edges) (whole graphs of <= 64 nodes)
int backward_dgrad_graph(egnn_ctx* c, hipStream_t st, int layer, const float* x, int e_first, int n_edges, const void* g_a2x,
                         const void* g_a2m, void* G, float* cd_x, float* cd_m, float* gd2_part);
}  // namespace egnn
