// Shared declarations of libegnn_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "../../include/egnn_amd.h"
#include "host_logic.h"   // set_error, padded model dimensions, argument checks, GEMM plan: the HIP-free part of the host side

namespace egnn {

constexpr int kThreads = 256;   // 4 wave64 per workgroup, one per SIMD
constexpr int kWaves = 4;

#define EGNN_HIP(call)                                                                      \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      egnn::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return EGNN_EHIP;                                                                     \
    }                                                                                       \
  } while (0)

// Packed parameters of one EGCL layer (all device memory, owned by the context).
struct LayerPack {
  bool packed = false;
  float* w1catT = nullptr;   // [H][TC]  transposed first-layer weights, columns = {Px|Qx|Pm|Qm}
  float* b1cat = nullptr;    // [TC]     first-layer bias (P columns only)
  float* wdx = nullptr;      // [WxP]    d^2 column of mlp_x.0
  float* wdm = nullptr;      // [WmP]    d^2 column of mlp_m.0
  float* w2x_f32 = nullptr;  // mlp_x.2 as f32 MFMA B fragments  [NB][KS4][64][4]
  void* w2x_bf16 = nullptr;  // mlp_x.2 as bf16 MFMA B fragments [NB][KS][64][8]
  float* b2x = nullptr;      // [WxP]
  float* w3x = nullptr;      // [WxP]    mlp_x.4 weight
  float* w2m_f32 = nullptr;  // mlp_m.2 fragments, N = MP, K = WmP
  void* w2m_bf16 = nullptr;
  float* b2m = nullptr;      // [MP]
  float* wa = nullptr;       // [MP]     attention.0 weight
  float* scal = nullptr;     // [4]      {mlp_x.4 bias, attention.0 bias}
  float* w1h_f32 = nullptr;  // mlp_h.0 fragments, N = WhP, K = K1P (= pad8(H+MP))
  float* b1h = nullptr;      // [WhP]
  float* w2h_f32 = nullptr;  // mlp_h.2 fragments, N = HP, K = WhP
  float* b2h = nullptr;      // [HP]
  // bf16 fast path: SiLU is evaluated as t * rcp(1 + exp2(t)) on t = -log2(e) * z.  The first-layer table,
  // biases and d^2 columns are pre-multiplied by -log2(e) and the following weights by -1/log2(e), which
  // removes one multiply per SiLU.
  float* sc = nullptr;       // [w1catT_s | b1cat_s | wdx_s | wdm_s | b2x_s | w3x_s | b2m_s | wa_s]
  void* w2x_bf16s = nullptr;
  void* w2m_bf16s = nullptr;
  void* w2x_bf16s_lo = nullptr;   // bf16 remainders of the scaled second-layer weights (precision bf16x3)
  void* w2m_bf16s_lo = nullptr;
  void* w2x_bf16s16 = nullptr;  // mlp_x.2 scaled, as v_mfma_f32_16x16x32_bf16 B fragments [N/16][K/32][64][8]
  void* w2m_bf16s16 = nullptr;  // mlp_m.2 scaled, same 16-column layout (edge_small.hip)
  void* w2xT_bf16 = nullptr;  // mlp_x.2 TRANSPOSED bf16 fragments for the backward dgrad (k = output n, column = hidden k)
  void* w2mT_bf16 = nullptr;  // mlp_m.2 transposed (K = MP, N = WmP)
  void* w1hl_bf16 = nullptr;  // scaled first layers as bf16 hi/lo B fragments [TC/32][3][hi|lo][64][8] (node_pre_hilo_kernel)
  void* w1h_bf16 = nullptr;   // mlp_h.0 bf16 fragments (N = WhP, K = K1Q)
  void* w2h_bf16p = nullptr;  // mlp_h.2 bf16 fragments, k in accumulator-row order
  // precision fp16: the streams of the bf16 path as fp16 fragments, every one multiplied by kF16WScale = 2^8 (kernels.h)
  void* w2x_f16s16 = nullptr;  // mlp_x.2 scaled, v_mfma_f32_16x16x32_f16 B fragments
  void* w2m_f16s = nullptr;    // mlp_m.2 scaled, v_mfma_f32_32x32x16_f16 B fragments
  void* w2m_f16s16 = nullptr;  // mlp_m.2 scaled, v_mfma_f32_16x16x32_f16 B fragments (edge_small.hip)
  void* w1h_f16 = nullptr;     // mlp_h.0
  void* w2h_f16p = nullptr;    // mlp_h.2, k in accumulator-row order
  // split-operand node MLP (node_post_bf16_kernel<., f16x8, true>): mlp_h.0 head / remainder with K padded to its ring's two
  // turns, mlp_h.2 remainder (its head is w2h_f16p); null when the shape is outside that kernel
  void *w1h_f16k = nullptr, *w1h_f16k_lo = nullptr, *w2h_f16p_lo = nullptr;
  // precision f16c8 (edge_f16c8.hip): e4m3 fragments [N/16][K/64][2][64][16 B] of the heads and remainders of the fp16 streams
  // above (hi | lo K blocks interleaved), and the e8m0 bytes of their block scales {x: hi, lo, m: hi, lo}
  void *w2x_c8 = nullptr, *w2m_c8 = nullptr;
  int* c8_exp = nullptr;
  // the same on 32x32 matrix tiles (edge_f16c8w.hip): mlp_x.2 as v_mfma_f32_32x32x16_f16 B fragments (mlp_m.2: w2m_f16s above) and
  // e4m3 fragments [N/32][K/32][2][64][16 B]
  void *w2x_f16s = nullptr, *w2x_c8w = nullptr, *w2m_c8w = nullptr;
};

constexpr int kGraphSteps = 8;   // reverse steps captured per hipGraph
struct Sampler {
  bool ready = false;
  int T = 0, A = 0, t = 0;
  int x_only = 0;                  // 1: positions diffuse, atom types fixed (test.py:253-279)
  float onehot_scale = 1.f;
  uint64_t seed = 0;
  const float* d_table = nullptr;  // [(T+1)*4] caller-owned
  float* pos = nullptr;            // [N][3]
  float* h = nullptr;              // [N][H]
  float* h_out = nullptr;          // [N][H]
  float* x_out = nullptr;          // [N][3]
  float* cond = nullptr;           // [N][H-A-1] constant conditioning block
  int* t_dev = nullptr;            // current t on device (graph replay reads it)
  int* bad = nullptr;              // [B] sticky non-finite flags
  hipGraph_t graph[2] = {nullptr, nullptr};            // [0]: one reverse step, [1]: kGraphSteps steps
  hipGraphExec_t graph_exec[2] = {nullptr, nullptr};
  int graph_prec = -1, graph_norm = -1;
};

}  // namespace egnn

struct egnn_ctx {
  int device = 0;
  // model
  int L = 0, H = 0, M = 0, Wm = 0, Wx = 0, Wh = 0;
  int WxP = 0, WmP = 0, MP = 0, WhP = 0, HP = 0, K1P = 0, K1Q = 0, TC = 0;
  int cbx = 0, cbm = 0;  // 32-column blocks per wave for the x / m second-layer GEMMs
  std::vector<egnn::LayerPack> layers;
  // graph
  int N = 0, E = 0, B = 0;
  const int32_t *edge_dst = nullptr, *edge_src = nullptr, *row_ptr = nullptr, *graph_ptr = nullptr,
                *node_graph = nullptr;
  // scratch (grown on demand by reserve())
  size_t cap_nodes = 0, cap_tiles = 0, cap_graphs = 0;
  float* table = nullptr;    // [N][TC] first-layer partial pre-activations
  float* agg_m = nullptr;    // [N][MP]
  float* agg_x = nullptr;    // [N][4]
  float* part_m = nullptr;   // [tiles][2][MP]
  float* part_x = nullptr;   // [tiles][2][4]
  float* node_d2 = nullptr;  // [N]
  float* gscale = nullptr;   // [B] sum of d^2 per graph (G^2); node_post applies 1/(G+1)
  int last_R = 64, last_nsplit_x = 1, last_path = 1;   // edge path chosen by the last launch_layer_begin
  bool sq_from_agg = false;             // node_post takes the d^2 sums from the coordinate sums' component 3
  bool small_ok = false;                // the partial slots were sized for 32-edge tiles (E <= 65536): edge_small.hip may run
  // A layer's hidden-split node_post leaves its 8 partial h' in h_partial; inside a multi-layer call the NEXT layer's node_pre
  // adds them up (and writes h') instead of a launch of its own (small graphs: every launch is a serial link of the step)
  struct { bool active = false; int hs = 0; const float* b2h = nullptr; float* h_out = nullptr; } pend;
  float* h_partial = nullptr;  // [8][N][H] partial node-MLP outputs (hidden-split node_post at small N)
  float* bwd_s = nullptr;    // [nsplit][chunk edges] column-split shares of s_e (backward recompute)
  size_t cap_bwd_s = 0;
  // egcl_forward_save: where the running layer's edge kernels leave what the backward needs (null = plain forward)
  void *save_s1x = nullptr, *save_s1m = nullptr, *save_t2x = nullptr, *save_t2m = nullptr;
  float* save_s = nullptr;   // [nsplit][E] column-split shares of s_e
  unsigned long long* stamps = nullptr;  // [2 kernels][8 waves][32 chunks][4] diagnostic time stamps
  float* h_tmp[2] = {nullptr, nullptr};  // [N][H] ping-pong between layers
  float* x_tmp[2] = {nullptr, nullptr};  // [N][3]
  egnn::Sampler smp;
  hipStream_t side = nullptr;                       // small graphs: the message edge kernel runs beside the coordinate kernel
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // profiling
  bool prof = false;
  std::vector<hipEvent_t> ev;  // pairs
  std::vector<int> ev_kind;    // 0 = edge kernel, 1 = node kernels
  size_t ev_used = 0;
};

namespace egnn {
int reserve(egnn_ctx* c);
int launch_layer(egnn_ctx* c, hipStream_t st, int layer, int prec, int norm_scope, const float* h,
                 const float* x, float* h_out, float* x_out, bool need_gscale = false, bool defer_finish = false);
int edge_rows_per_tile(int prec);
// backward recompute on the forward's bf16 edge kernels (egcl_backward_edge_recompute)
int backward_recompute_supported(egnn_ctx* c);
int backward_table(egnn_ctx* c, hipStream_t st, int layer, const float* h);
int backward_recompute(egnn_ctx* c, hipStream_t st, int layer, const float* x, const float* g_sum_x, const float* g_sum_m,
                       int e_first, int n_edges, void* s1x, void* s1m, void* g_a2x, void* g_a2m, float* s_halves,
                       float* g_b2x, float* g_w3, float* g_b3, float* g_b2m, float* g_wa, float* g_ba);
int backward_dgrad(egnn_ctx* c, hipStream_t st, int layer, const float* x, int e_first, int n_edges, const void* g_a2x,
                   const void* g_a2m, void* g_a1x, void* g_a1m);
int backward_heads_saved(egnn_ctx* c, hipStream_t st, int layer, const float* x, const float* g_sum_x, const float* g_sum_m,
                         int e_first, int n_edges, void* t2x, void* t2m, float* g_b2x, float* g_w3, float* g_b3, float* g_b2m,
                         float* g_wa, float* g_ba);
bool heads_saved_supported(int WxP, int MP);
int launch_heads_saved(int n_edges, const int* dst, const int* src, const float* x, const float* g_sum_x, const float* g_sum_m,
                       int WxP, int MP, const float* w3s, const float* was, const float* scal, void* t2x, void* t2m,
                       float* g_b2x, float* g_w3, float* g_b3, float* g_b2m, float* g_wa, float* g_ba, hipStream_t st);
int init_edge_dgrad_attributes();
int fork_streams(egnn_ctx* c);
int launch_edge_dgrad(int N, int E, const int* dst, const int* src, const float* x, const void* table, int TC, int offP, int offQ,
                      const float* wd, const void* g_a2, int Kd, const void* w2t, int KP, void* g_a1_out, hipStream_t st);
}  // namespace egnn
