// Host-only logic of libegnn_amd: everything that decides, validates or tabulates WITHOUT touching the HIP runtime.
// Kept in one plain C++ translation unit (host_logic.cpp) so that it also builds with g++ -fsanitize=address,undefined into
// a CPU-only library the build container's tests run (make asan -> build/libegnn_host_asan.so; tests/test_host_asan.py):
// the schedule builder, argument / shape validation of every entry point that has any, the padded model dimensions,
// the split-K plan of the weight-gradient GEMM and the small-graph fork decision.  No hip header is included here.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/egnn_amd.h"

namespace egnn {

void set_error(const char* fmt, ...);

// Padded dimensions of one model (egnn_set_model): zero-padded widths contribute SiLU(0) * 0 = 0.
struct ModelDims {
  int WxP, WmP, MP, WhP, HP, K1P, K1Q, TC, cbx, cbm;
};
constexpr int kMaxCB = 8;       // 32-column blocks per wave in the fused edge GEMMs (N <= 1024)
constexpr int kPostMaxOB = 8;   // output column blocks of node_post (H <= 256)
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
// EGNN_OK and *out filled, or EGNN_EINVAL with the error text set; lds_ok(K1P, WhP, MP) = the caller's LDS-budget predicate
int model_dims(int L, int H, int M, int Wm, int Wx, int Wh, ModelDims* out);

int graph_args_check(bool have_model, int N, int E, int B, const void* edge_dst, const void* edge_src, const void* row_ptr,
                     const void* graph_ptr, const void* node_graph);
int precision_scope_check(bool ready, int prec, int norm_scope);

// split-K plan of egnn_gemm_tn_bf16 (gemm_tn.hip): column tile BN, tiles along N, number of slices S, k-steps per slice
constexpr int kGemmBM = 256, kGemmBK = 32;
void plan_gemm_tn(int E, int M, int N, int& BN, int& tiles_n, int& S, int& steps_per_slice);
int gemm_tn_args_check(int E, int M, int N, const void* A, int lda, const void* B, int ldb, const void* C, int ldc, int rows,
                       int cols, const void* workspace, size_t workspace_bytes);
int gemm_rows_args_check(int E, const void* A0, int lda0, int K0, const void* W0, const void* A1, int lda1, int K1, const void* W1,
                         const void* out, int ldo);
int dense_rows_args_check(int N, int K, int J, const void* in, const void* W, const void* b, const void* out);

// the message edge kernel goes to the caller's side stream when the coordinate kernel's last round of workgroups leaves
// enough CUs idle for all message workgroups (two per CU)
bool fork_candidate(int E, int WxP);

}  // namespace egnn
