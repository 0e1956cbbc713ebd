// EGNN denoiser forward path for MI355X (gfx950 / CDNA4).
//
// One EGCL layer (reference: EquivariantGraphNeuralNetwork.py:55-71) is three stages:
//
// (bf16 default path: 4 launches per layer -- node_pre, coordinate edge kernel, message edge kernel, node_post; the
//  normaliser's d^2 sums ride in component 3 of the coordinate sums.  Other paths add node_d2 + graph_sum.)
//   node_pre   per node: first-layer partial pre-activations of BOTH edge MLPs.  The reference
//              feeds [h_i | h_j | d^2] (2H+1 wide) through Linear(2H+1, W) per EDGE (:56-57,:63-64);
//              because that layer is linear in its input, W1.[h_i|h_j|d2] + b =
//              (W1[:, :H] h_i + b) + (W1[:, H:2H] h_j) + W1[:, 2H] d2, so it is evaluated once per
//              NODE into a table {Px|Qx|Pm|Qm} and the per-edge work is two row reads and one fma.
//   edge       fused per-edge chain on a tile of R edges (CSR order, sorted by receiving node):
//              gather (x_i, x_j, table rows) -> d^2 -> SiLU -> second-layer GEMMs on MFMA with the
//              [R, W] hidden activations living only in LDS/registers -> SiLU -> mlp_x.4 dot /
//              attention gate -> segment sums per receiving node.  No [E, *] tensor reaches HBM.
//   node_post  per node: gather the segment sums, mlp_h on MFMA (hidden never leaves the CU),
//              coordinate update x' = x + sum_x / (G + 1).
//
// Precision: EGNN_PREC_F32 uses v_mfma_f32_32x32x2_f32 (exact fp32 fma chains) everywhere.  EGNN_PREC_BF16 uses
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation for the two big per-edge GEMMs AND for the node MLP mlp_h
// (node_post_bf16_kernel, csrc/node_bf16.hip), and a half-precision first-layer table; geometry (d^2, coordinate
// differences), SiLU / sigmoid, all segment sums and the first-layer table's arithmetic stay fp32 in both.
#include <stdarg.h>
#include <stdlib.h>

#include "bwd_graph.h"
#include "common.h"
#include "kernels.h"

namespace egnn {

// ------------------------------------------------------------------------------------------------
// parameter packing
// ------------------------------------------------------------------------------------------------
// B fragments of D = A.B with B[k][n] = W[n][k] (nn.Linear weight [Nout, K], leading dim ldw,
// column offset koff) for v_mfma_f32_32x32x2_f32: lane l holds B[k = 2*ks + (l>>5)][n = 32*nb + (l&31)].
// Four consecutive k-steps are stored together so one 16-byte load per lane feeds 4 MFMAs:
// out[((nb*KS4 + ks4)*64 + lane)*4 + s] = W[32nb + (l&31)][8*ks4 + 2*s + (l>>5)].
__global__ void pack_frags_f32(const float* __restrict__ W, int Nout, int K, int ldw, int NP, int KP,
                               float* __restrict__ out) {
  const int KS4 = KP / 8;
  const size_t total = (size_t)(NP / 32) * KS4 * 64 * 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int s = i & 3, lane = (i >> 2) & 63;
    const size_t f = i >> 8;
    const int ks4 = f % KS4, nb = f / KS4;
    const int n = 32 * nb + (lane & 31), k = 8 * ks4 + 2 * s + (lane >> 5);
    out[i] = (n < Nout && k < K) ? W[(size_t)n * ldw + k] : 0.f;
  }
}
// v_mfma_f32_32x32x16_bf16: lane l holds B[k = 16*ks + 8*(l>>5) + j][n = 32*nb + (l&31)], j = 0..7.
// out[((nb*KS + ks)*64 + lane)*8 + j]
// OT = __bf16, or _Float16 for precision fp16 (scale then carries kF16WScale; clamped to the finite fp16 range)
// LO: the remainder v - OT(v) of the same element (split-operand products)
template <typename OT, bool LO = false>
__device__ __forceinline__ OT to_operand(float v) {
  if constexpr (sizeof(OT) == 2 && !__is_same(OT, __bf16)) v = fminf(fmaxf(v, -65504.f), 65504.f);
  if constexpr (LO) return (OT)(v - (float)(OT)v);
  return (OT)v;
}
template <typename OT, bool LO = false>
__global__ void pack_frags_bf16(const float* __restrict__ W, int Nout, int K, int ldw, int NP, int KP,
                                OT* __restrict__ out, float scale) {
  const int KS = KP / 16;
  const size_t total = (size_t)(NP / 32) * KS * 64 * 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63;
    const size_t f = i >> 9;
    const int ks = f % KS, nb = f / KS;
    const int n = 32 * nb + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
    out[i] = to_operand<OT, LO>((n < Nout && k < K) ? W[(size_t)n * ldw + k] * scale : 0.f);
  }
}
// v_mfma_f32_16x16x32_bf16: lane l holds B[k = 32*ks + 8*(l>>4) + j][n = 16*nb + (l&15)], j = 0..7.
// out[((nb*KS + ks)*64 + lane)*8 + j]
template <typename OT>
__global__ void pack_frags_bf16_n16(const float* __restrict__ W, int Nout, int K, int ldw, int NP, int KP,
                                    OT* __restrict__ out, float scale) {
  const int KS = KP / 32;
  const size_t total = (size_t)(NP / 16) * KS * 64 * 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63;
    const size_t f = i >> 9;
    const int ks = f % KS, nb = f / KS;
    const int n = 16 * nb + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
    out[i] = to_operand<OT>((n < Nout && k < K) ? W[(size_t)n * ldw + k] * scale : 0.f);
  }
}
// bf16 remainder of the same fragments: out = bf16(v - bf16(v)), v = W * scale (precision bf16x3)
__global__ void pack_frags_bf16_lo(const float* __restrict__ W, int Nout, int K, int ldw, int NP, int KP,
                                   __bf16* __restrict__ out, float scale) {
  const int KS = KP / 16;
  const size_t total = (size_t)(NP / 32) * KS * 64 * 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63;
    const size_t f = i >> 9;
    const int ks = f % KS, nb = f / KS;
    const int n = 32 * nb + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
    const float v = (n < Nout && k < K) ? W[(size_t)n * ldw + k] * scale : 0.f;
    out[i] = (__bf16)(v - (float)(__bf16)v);
  }
}
// mlp_h.2 as the A operand of out^T = W2h . hidden^T where hidden^T comes straight from an accumulator tile:
// element j of lane half hh in k-step ks is hidden unit 32*(ks/2) + 16*(ks%2) + 8*(j>>2) + 4*hh + (j&3).
template <typename OT, bool LO = false>
__global__ void pack_frags_bf16_accperm(const float* __restrict__ W, int Nout, int K, int ldw, int NP, int KP,
                                        OT* __restrict__ out, float scale) {
  const int KS = KP / 16;
  const size_t total = (size_t)(NP / 32) * KS * 64 * 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63;
    const size_t f = i >> 9;
    const int ks = f % KS, nb = f / KS;
    const int n = 32 * nb + (lane & 31);
    const int k = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
    out[i] = to_operand<OT, LO>((n < Nout && k < K) ? W[(size_t)n * ldw + k] * scale : 0.f);
  }
}
// the same fragment layout for B[k][n] = W[k][n] (the transposed use of an nn.Linear weight: dgrad g . W)
__global__ void pack_frags_bf16_T(const float* __restrict__ W, int Krows, int Ncols, int ldw, int NP, int KP,
                                  __bf16* __restrict__ out) {
  const int KS = KP / 16;
  const size_t total = (size_t)(NP / 32) * KS * 64 * 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63;
    const size_t f = i >> 9;
    const int ks = f % KS, nb = f / KS;
    const int n = 32 * nb + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
    out[i] = (__bf16)((n < Ncols && k < Krows) ? W[(size_t)k * ldw + n] : 0.f);
  }
}
__global__ void scale_copy(const float* __restrict__ src, size_t n, float scale, float* __restrict__ dst) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i] * scale;
}
__global__ void pad_copy(const float* __restrict__ src, int n, int stride, float* __restrict__ dst, int nP) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nP; i += gridDim.x * blockDim.x)
    dst[i] = i < n ? src[(size_t)i * stride] : 0.f;
}
// first layers of mlp_x / mlp_m, split per input block and transposed: w1catT[h][col]
__global__ void pack_first(const float* __restrict__ x0_w, const float* __restrict__ x0_b,
                           const float* __restrict__ m0_w, const float* __restrict__ m0_b, int H, int Wx,
                           int Wm, int WxP, int WmP, float* __restrict__ w1catT, float* __restrict__ b1cat) {
  const int TC = 2 * WxP + 2 * WmP, ld = 2 * H + 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < TC * (H + 1); i += gridDim.x * blockDim.x) {
    const int col = i % TC, h = i / TC;  // h == H -> bias row
    int k, W, hoff;
    const float *w, *b;
    if (col < WxP) { k = col; W = Wx; hoff = 0; w = x0_w; b = x0_b; }
    else if (col < 2 * WxP) { k = col - WxP; W = Wx; hoff = H; w = x0_w; b = nullptr; }
    else if (col < 2 * WxP + WmP) { k = col - 2 * WxP; W = Wm; hoff = 0; w = m0_w; b = m0_b; }
    else { k = col - 2 * WxP - WmP; W = Wm; hoff = H; w = m0_w; b = nullptr; }
    if (h < H) w1catT[(size_t)h * TC + col] = k < W ? w[(size_t)k * ld + hoff + h] : 0.f;
    else b1cat[col] = (k < W && b) ? b[k] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// node_pre: table[n][col] = b1cat[col] + sum_h h[n][h] * w1catT[h][col]
// ------------------------------------------------------------------------------------------------
constexpr int kPreNodes = 16;
// fp16 table entries saturate at +-32000 instead of overflowing: the edge kernel adds two of them in fp16, and an
// infinite pre-activation would turn SiLU's t * rcp(1 + exp2(t)) into inf * 0.  (Scaled pre-activations of that
// size only occur once a reverse chain has already diverged.)
__device__ __forceinline__ void table_store(float* p, float v) { *p = v; }
__device__ __forceinline__ void table_store(_Float16* p, float v) { *p = (_Float16)fminf(fmaxf(v, -32000.f), 32000.f); }

template <typename TT>
__global__ __launch_bounds__(kThreads) void node_pre_kernel(const float* __restrict__ h, int N, int H,
                                                            const float* __restrict__ w1catT,
                                                            const float* __restrict__ b1cat, int TC,
                                                            TT* __restrict__ table) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* hs = reinterpret_cast<float*>(smem_raw);  // [kPreNodes][H]
  const int n0 = blockIdx.x * kPreNodes;
  for (int i = threadIdx.x; i < kPreNodes * H; i += kThreads) {
    const int n = n0 + i / H;
    hs[i] = n < N ? h[(size_t)n * H + i % H] : 0.f;
  }
  __syncthreads();
  const int col = blockIdx.y * kThreads + threadIdx.x;
  if (col >= TC) return;
  float acc[kPreNodes];
  const float b = b1cat[col];
#pragma unroll
  for (int i = 0; i < kPreNodes; ++i) acc[i] = b;
  for (int k = 0; k < H; ++k) {
    const float w = w1catT[(size_t)k * TC + col];
#pragma unroll
    for (int i = 0; i < kPreNodes; ++i) acc[i] = fmaf(hs[i * H + k], w, acc[i]);
  }
#pragma unroll
  for (int i = 0; i < kPreNodes; ++i)
    if (n0 + i < N) table_store(table + (size_t)(n0 + i) * TC + col, acc[i]);
}

// The same table on the matrix cores: exact fp32 v_mfma_f32_32x32x2_f32 (K = H is tiny, the kernel is bound by
// the table write: 268 MB fp32 / 134 MB fp16 at C2).  Workgroup = 32 nodes x 512 columns, wave w owns 4 column blocks; A = h tile from LDS,
// B = w1catT rows straight from L2 (128 B per half-wave, coalesced), accumulator initialised with the bias.
constexpr int kPre2Nodes = 32, kPre2Cols = 512;
template <typename TT>
__global__ __launch_bounds__(kThreads) void node_pre_mfma_kernel(const float* __restrict__ h, int N, int H,
                                                                 const float* __restrict__ w1catT,
                                                                 const float* __restrict__ b1cat, int TC,
                                                                 TT* __restrict__ table) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* hs = reinterpret_cast<float*>(smem_raw);  // [HP2][33], HP2 = H rounded up to even
  const int HP2 = (H + 1) & ~1;
  const int n0 = blockIdx.x * kPre2Nodes;
  for (int i = threadIdx.x; i < kPre2Nodes * HP2; i += kThreads) {
    const int node = i / HP2, k = i % HP2, n = n0 + node;
    hs[k * 33 + node] = (n < N && k < H) ? h[(size_t)n * H + k] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
  const int col0 = blockIdx.y * kPre2Cols + wave * 128;
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = col0 + 32 * j + r;
    const float b = col < TC ? b1cat[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = b;
  }
  // all B values of this lane (KS k-steps x 4 column blocks) are requested up front: one L2 round trip
  constexpr int kMaxKS = 32;   // H <= 64 (checked on the host)
  const int KS = HP2 / 2;
  float bw[4][kMaxKS];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = col0 + 32 * j + r;
#pragma unroll
    for (int s = 0; s < kMaxKS; ++s) {
      const int k = 2 * s + hh;
      bw[j][s] = (s < KS && k < H && col < TC) ? w1catT[(size_t)k * TC + col] : 0.f;
    }
  }
#pragma unroll
  for (int s = 0; s < kMaxKS; ++s)
    if (s < KS) {
      const float a = hs[(2 * s + hh) * 33 + r];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[j][s], acc[j], 0, 0, 0);
    }
  if constexpr (sizeof(TT) == 2) {
    // fp16 table: a lane's 2-byte column of 16 rows would touch 16 rows x 64 B per half-wave; stage the 32 x 512
    // tile in LDS and store whole 16-byte pieces, 1 KiB contiguous per row
    constexpr int kLd = kPre2Cols + 8;   // halves; the pad de-phases the two row groups of a wave
    __syncthreads();                     // every wave is done reading the h tile
    TT* stg = reinterpret_cast<TT*>(smem_raw);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) table_store(stg + acc_row(i, lane) * kLd + wave * 128 + 32 * j + r, acc[j][i]);
    __syncthreads();
    const int cbase = blockIdx.y * kPre2Cols;
    for (int q = threadIdx.x; q < kPre2Nodes * (kPre2Cols / 8); q += kThreads) {
      const int row = q / (kPre2Cols / 8), piece = q % (kPre2Cols / 8), n = n0 + row, col = cbase + 8 * piece;
      if (n < N && col < TC)
        *reinterpret_cast<f32x4*>(table + (size_t)n * TC + col) = *reinterpret_cast<const f32x4*>(stg + row * kLd + 8 * piece);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + 32 * j + r;
      if (col < TC) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int n = n0 + acc_row(i, lane);
          if (n < N) table_store(table + (size_t)n * TC + col, acc[j][i]);
        }
      }
    }
  }
}

// ---- node_pre for the half-precision table on bf16 MFMA with a hi/lo split -------------------------------------------
// The f32-input MFMA runs at 1/16 of the bf16 rate and made node_pre MFMA-bound at ~3x its HBM-write time.  Here both
// operands are split h = h_hi + h_lo, W = W_hi + W_lo (bf16 each, lo = bf16(v - hi)) and the product is
// h_hi.W_hi + h_hi.W_lo + h_lo.W_hi: three v_mfma_f32_32x32x16_bf16 per 16-deep k-step with fp32 accumulation.  The dropped
// h_lo.W_lo term and the 16-bit representation bound the error at ~2^-16 relative -- 30x below the half-precision rounding
// of the table entry itself (2^-11); K = H <= 48 is 3 k-steps, so a 32 x 32 tile costs 9 x 32 = 288 MFMA cycles
// instead of 18 x 64 = 1152.
// B fragments of the (pre-scaled) first-layer weights: [TC/32 column blocks][3 k-steps][hi|lo][64 lanes][8 bf16].
__global__ void pack_w1_hilo(const float* __restrict__ w1catT, int H, int TC, __bf16* __restrict__ out) {
  const size_t total = (size_t)(TC / 32) * 3 * 64 * 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63;
    const size_t f = i >> 9;
    const int ks = f % 3, nb = f / 3;
    const int n = 32 * nb + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
    const float v = k < H ? w1catT[(size_t)k * TC + n] : 0.f;
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    const size_t base = (f * 2) * 512 + (size_t)lane * 8 + j;
    out[base] = hi;
    out[base + 512] = lo;
  }
}

// workgroup = 32 nodes x (4 waves x NBW column blocks): NBW = 8 (1024 columns) to amortise the h tile at large N,
// NBW = 2 (256 columns) when the whole launch is a handful of workgroups and their serial length is the layer's latency
constexpr int kPre3Nodes = 32;
// fin_hs > 0: h is not there yet -- the previous layer's hidden-split node_post left fin_hs partial sums in fin_partial
// [fin_hs][N][H]; this kernel adds them up in split order (+ the bias fin_b2h: what node_post_finish_kernel does) and the
// workgroups of column block 0 write h' to h_write (the next node_post and the caller read it there).
template <int NBW>
__global__ __launch_bounds__(kThreads) void node_pre_hilo_kernel(const float* __restrict__ h, int N, int H,
                                                                 const bf16x8* __restrict__ w1hl,
                                                                 const float* __restrict__ b1cat, int TC,
                                                                 _Float16* __restrict__ table, int fin_hs,
                                                                 const float* __restrict__ fin_partial,
                                                                 const float* __restrict__ fin_b2h, float* __restrict__ h_write) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* hs = reinterpret_cast<float*>(smem_raw);   // [48][33] transposed h tile, zero-padded to K = 48
  const int n0 = blockIdx.x * kPre3Nodes;
  for (int i = threadIdx.x; i < kPre3Nodes * 48; i += kThreads) {
    const int node = i / 48, k = i % 48, n = n0 + node;
    float v = 0.f;
    if (n < N && k < H) {
      if (fin_hs > 0) {
        float pv[8];   // (the split is 8-fold: all loads first, then the sum in split order -- a load + add loop is 8 serial round trips)
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) pv[sp] = fin_partial[((size_t)(sp < fin_hs ? sp : 0) * N + n) * H + k];
        v = fin_b2h[k];
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) v += sp < fin_hs ? pv[sp] : 0.f;
        if (blockIdx.y == 0) h_write[(size_t)n * H + k] = v;
      } else v = h[(size_t)n * H + k];
    }
    hs[k * 33 + node] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
  // A fragments (node r, k = 16 ks + 8 hh + j), hi and lo parts
  bf16x8 ahi[3], alo[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = hs[(16 * ks + 8 * hh + j) * 33 + r];
      const __bf16 hi = (__bf16)v;
      ahi[ks][j] = hi;
      alo[ks][j] = (__bf16)(v - (float)hi);
    }
  const int cb0 = (blockIdx.y * 4 + wave) * NBW;   // first column block of this wave
  constexpr int kLd = 128 + 8;   // halves per staged row (up to 4 column blocks = 128 columns at a time)
  constexpr int kPairs = NBW >= 4 ? 2 : NBW / 2;   // pairs of column blocks per staging round
  _Float16* stg = reinterpret_cast<_Float16*>(smem_raw + 48 * 33 * 4) + (size_t)wave * 32 * kLd;
  for (int half8 = 0; half8 < (NBW + 3) / 4; ++half8) {
#pragma unroll
    for (int pair = 0; pair < kPairs; ++pair) {   // 2 column blocks = 12 fragment loads in flight
      bf16x8 bf[2][3][2];
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const int cb = cb0 + 4 * half8 + 2 * pair + jb;
        const bool ok = 32 * cb < TC;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
          for (int q = 0; q < 2; ++q) bf[jb][ks][q] = ok ? w1hl[(((size_t)cb * 3 + ks) * 2 + q) * 64 + lane] : bf16x8{};
      }
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const int cb = cb0 + 4 * half8 + 2 * pair + jb, col = 32 * cb + r;
        const float b = col < TC ? b1cat[col] : 0.f;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = b;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[ks], bf[jb][ks][0], acc, 0, 0, 0);   // small terms first
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[ks], bf[jb][ks][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[ks], bf[jb][ks][0], acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) table_store(stg + acc_row(i, lane) * kLd + 32 * (2 * pair + jb) + r, acc[i]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // this wave's 32 rows x (64 kPairs) columns as 16-byte pieces: up to 256 contiguous bytes per row
    const int cbase = 32 * (cb0 + 4 * half8);
    constexpr int kPieces = 8 * kPairs;   // 16-byte pieces per row
    for (int q = lane; q < 32 * kPieces; q += 64) {
      const int row = q / kPieces, piece = q % kPieces, n = n0 + row, col = cbase + 8 * piece;
      if (n < N && col < TC)
        *reinterpret_cast<f32x4*>(table + (size_t)n * TC + col) = *reinterpret_cast<const f32x4*>(stg + row * kLd + 8 * piece);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// node_d2[n] = sum over the edges received by n of |x_n - x_src|^2 (fixed order -> deterministic)
__global__ void node_d2_kernel(const float* __restrict__ x, const int* __restrict__ row_ptr,
                               const int* __restrict__ edge_src, int N, float* __restrict__ node_d2) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float x0 = x[3 * n], x1 = x[3 * n + 1], x2 = x[3 * n + 2];
  float s = 0.f;
  for (int e = row_ptr[n]; e < row_ptr[n + 1]; ++e) {
    const int j = edge_src[e];
    const float a = x0 - x[3 * j], b = x1 - x[3 * j + 1], c = x2 - x[3 * j + 2];
    s += a * a + b * b + c * c;
  }
  node_d2[n] = s;
}
// gsum[g] = sum_{n in g} node_d2[n] (= G^2 of the graph, or of the whole call in 'call' scope); node_post turns
// it into 1 / (G + 1).  One workgroup per graph, fixed summation order.
__global__ __launch_bounds__(kThreads) void graph_sum_kernel(const float* __restrict__ node_d2,
                                                               const int* __restrict__ graph_ptr, int N,
                                                               int per_graph, float* __restrict__ gscale) {
  __shared__ float red[kThreads];
  const int lo = per_graph ? graph_ptr[blockIdx.x] : 0;
  const int hi = per_graph ? graph_ptr[blockIdx.x + 1] : N;
  float s = 0.f;
  for (int n = lo + threadIdx.x; n < hi; n += kThreads) s += node_d2[n];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) gscale[blockIdx.x] = red[0];
}

// The same sums from what the v4 edge kernels leave in component 3 of the coordinate sums (no pass over the edges):
// only launched where the sums are needed OUTSIDE node_post -- 'call' scope, the two-stage (partitioned) layer and
// egcl_read_aggregates; in the per-graph sampling path node_post adds them up itself.
__global__ __launch_bounds__(kThreads) void graph_sq_sums_kernel(const int* __restrict__ row_ptr, int R,
                                                                   const float* __restrict__ agg_x,
                                                                   const float* __restrict__ part_x,
                                                                   const int* __restrict__ graph_ptr, int N, int per_graph,
                                                                   float* __restrict__ gscale) {
  __shared__ float red[kThreads];
  const int lo = per_graph ? graph_ptr[blockIdx.x] : 0;
  const int hi = per_graph ? graph_ptr[blockIdx.x + 1] : N;
  float s = 0.f;
  for (int n = lo + threadIdx.x; n < hi; n += kThreads) s += node_sq_sum(n, row_ptr, R, agg_x, part_x);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) gscale[blockIdx.x] = red[0];
}

// ------------------------------------------------------------------------------------------------
// fused edge kernel
// ------------------------------------------------------------------------------------------------
// ---- second-layer GEMM of one edge MLP over a tile of R = 32*RB edges --------------------------
// acc[rb][cb] (+)= SiLU(P[dst] + Q[src] + wd*d2)[R, KP] . W2^T[KP, wave's 32*CB columns]
// The hidden activation is produced KC columns at a time into a double-buffered LDS chunk that all
// four waves read as MFMA A fragments; each wave streams its own B fragments straight from the
// packed weights (one coalesced 1 KiB load per fragment).
template <int RB, int CB>
__device__ __forceinline__ void gemm_bf16(const float* __restrict__ table, int TC, int offP, int offQ,
                                          const float* __restrict__ wd, int KP,
                                          const bf16x8* __restrict__ w2, const int* s_dst,
                                          const int* s_src, const float* s_d2, char* s_a1,
                                          f32x16 (&acc)[RB][CB]) {
  constexpr int R = 32 * RB, RPAD = R + 1, KC = 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int NC = KP / KC, KS = KP / 16;
  const size_t bufBytes = edge_a1_bytes(R);
  const int kg = tid & 7, rsub = tid >> 3;  // 8 k-groups of 8 columns, 32 rows per pass

  const bf16x8* wb[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) wb[cb] = w2 + ((size_t)(wave * CB + cb) * KS) * 64 + lane;

  auto build = [&](int c, char* buf) {
    const int k0 = c * KC + kg * 8;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wd + k0), w1 = *reinterpret_cast<const f32x4*>(wd + k0 + 4);
#pragma unroll
    for (int pass = 0; pass < RB; ++pass) {
      const int row = pass * 32 + rsub;
      const float* pp = table + (size_t)s_dst[row] * TC + offP + k0;
      const float* qq = table + (size_t)s_src[row] * TC + offQ + k0;
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(pp), p1 = *reinterpret_cast<const f32x4*>(pp + 4);
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(qq), q1 = *reinterpret_cast<const f32x4*>(qq + 4);
      const float d2 = s_d2[row];
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (__bf16)silu_f(fmaf(w0[j], d2, p0[j] + q0[j]));
        o[j + 4] = (__bf16)silu_f(fmaf(w1[j], d2, p1[j] + q1[j]));
      }
      *reinterpret_cast<bf16x8*>(buf + ((size_t)kg * RPAD + row) * 16) = o;
    }
  };

  build(0, s_a1);
  __syncthreads();
  bf16x8 bcur[CB], bnxt[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) bcur[cb] = wb[cb][0];

  for (int c = 0; c < NC; ++c) {
    char* cur = s_a1 + (c & 1) * bufBytes;
    char* nxt = s_a1 + ((c + 1) & 1) * bufBytes;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ks = c * 4 + s;
      const int ksn = ks + 1 < KS ? ks + 1 : ks;  // last prefetch re-reads the final fragment (in bounds)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) bnxt[cb] = wb[cb][(size_t)ksn * 64];
      bf16x8 a[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
        a[rb] = *reinterpret_cast<const bf16x8*>(cur + ((size_t)(s * 2 + hh) * RPAD + 32 * rb + r) * 16);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rb], bcur[cb], acc[rb][cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) bcur[cb] = bnxt[cb];
    }
    if (c + 1 < NC) build(c + 1, nxt);
    __syncthreads();
  }
}

template <int RB, int CB>
__device__ __forceinline__ void gemm_f32(const float* __restrict__ table, int TC, int offP, int offQ,
                                         const float* __restrict__ wd, int KP,
                                         const f32x4* __restrict__ w2, const int* s_dst, const int* s_src,
                                         const float* s_d2, char* s_a1, f32x16 (&acc)[RB][CB]) {
  constexpr int R = 32 * RB, KC = 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int NC = KP / KC, KS4 = KP / 8;
  const size_t bufBytes = edge_a1_bytes(R);
  const int kg = tid & 3, rsub = tid >> 2;  // 4 k-groups of 8 columns, 64 rows per pass

  const f32x4* wb[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) wb[cb] = w2 + ((size_t)(wave * CB + cb) * KS4) * 64 + lane;

  auto build = [&](int c, float* buf) {
    const int k0 = c * KC + kg * 8;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wd + k0), w1 = *reinterpret_cast<const f32x4*>(wd + k0 + 4);
#pragma unroll
    for (int pass = 0; pass < (R + 63) / 64; ++pass) {
      const int row = pass * 64 + rsub;
      if (row < R) {
        const float* pp = table + (size_t)s_dst[row] * TC + offP + k0;
        const float* qq = table + (size_t)s_src[row] * TC + offQ + k0;
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(pp), p1 = *reinterpret_cast<const f32x4*>(pp + 4);
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(qq), q1 = *reinterpret_cast<const f32x4*>(qq + 4);
        const float d2 = s_d2[row];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          buf[(kg * 8 + j) * R + row] = silu_f(fmaf(w0[j], d2, p0[j] + q0[j]));
          buf[(kg * 8 + 4 + j) * R + row] = silu_f(fmaf(w1[j], d2, p1[j] + q1[j]));
        }
      }
    }
  };

  build(0, reinterpret_cast<float*>(s_a1));
  __syncthreads();
  for (int c = 0; c < NC; ++c) {
    const float* cur = reinterpret_cast<const float*>(s_a1 + (c & 1) * bufBytes);
    float* nxt = reinterpret_cast<float*>(s_a1 + ((c + 1) & 1) * bufBytes);
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // 4 groups of 4 k-steps (k-step = 2 columns)
      f32x4 b[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) b[cb] = wb[cb][(size_t)(c * 4 + g) * 64];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float a[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) a[rb] = cur[(8 * g + 2 * s + hh) * R + 32 * rb + r];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb], b[cb][s], acc[rb][cb], 0, 0, 0);
      }
    }
    if (c + 1 < NC) build(c + 1, nxt);
    __syncthreads();
  }
}

template <int PREC, int RB, int CB>
__device__ __forceinline__ void run_gemm(const EdgeParams& p, int offP, int offQ, const float* wd, int KP,
                                         const void* w2, const int* s_dst, const int* s_src,
                                         const float* s_d2, char* s_a1, f32x16 (&acc)[RB][CB]) {
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;
  if constexpr (PREC == EGNN_PREC_BF16)
    gemm_bf16<RB, CB>(p.table, p.TC, offP, offQ, wd, KP, reinterpret_cast<const bf16x8*>(w2), s_dst, s_src,
                      s_d2, s_a1, acc);
  else
    gemm_f32<RB, CB>(p.table, p.TC, offP, offQ, wd, KP, reinterpret_cast<const f32x4*>(w2), s_dst, s_src,
                     s_d2, s_a1, acc);
}

// mlp_x: s[row] = b3 + sum_n w3[n] * SiLU(acc[row][n] + b2[n])      (EquivariantGraphNeuralNetwork.py:19-25)
template <int PREC, int RB, int CB>
__device__ __forceinline__ void phase_x(const EdgeParams& p, const int* s_dst, const int* s_src,
                                     const float* s_d2, char* s_a1, float* s_part) {
  constexpr int R = 32 * RB;
  f32x16 acc[RB][CB];
  run_gemm<PREC, RB, CB>(p, 0, p.WxP, p.wdx, p.WxP, p.w2x, s_dst, s_src, s_d2, s_a1, acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float part[RB][16];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) part[rb][i] = 0.f;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int n = 32 * (wave * CB + cb) + (lane & 31);
    const float b = p.b2x[n], w = p.w3x[n];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) part[rb][i] = fmaf(w, silu_f(acc[rb][cb][i] + b), part[rb][i]);
  }
  // sum over the 32 columns held by the lanes of each half-wave
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = part[rb][i];
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
      if ((lane & 31) == 0) s_part[wave * R + 32 * rb + acc_row(i, lane)] = v;
    }
}

// mlp_m: msg[row][n] = SiLU(acc[row][n] + b2[n])                     (:13-18), gate applied later
template <int PREC, int RB, int CB>
__device__ __forceinline__ void phase_m(const EdgeParams& p, const int* s_dst, const int* s_src,
                                     const float* s_d2, char* s_a1, float* s_msg) {
  f32x16 acc[RB][CB];
  run_gemm<PREC, RB, CB>(p, 2 * p.WxP, 2 * p.WxP + p.WmP, p.wdm, p.WmP, p.w2m, s_dst, s_src, s_d2, s_a1, acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ld = p.MP + 1;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int n = 32 * (wave * CB + cb) + (lane & 31);
    const float b = p.b2m[n];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        s_msg[(32 * rb + acc_row(i, lane)) * ld + n] = silu_f(acc[rb][cb][i] + b);
  }
}

template <int PREC, int RB>
__global__ __launch_bounds__(kThreads, 1) void edge_kernel(const EdgeParams p) {
  constexpr int R = 32 * RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_dst = reinterpret_cast<int*>(smem);
  int* s_src = s_dst + R;
  float* s_d2 = reinterpret_cast<float*>(s_src + R);
  float* s_diff = s_d2 + R;      // [3][R]
  float* s_sval = s_diff + 3 * R;
  float* s_gate = s_sval + R;
  float* s_part = s_gate + R;    // [4][R]
  char* s_a1 = smem + edge_smem_small(R);
  float* s_msg = reinterpret_cast<float*>(s_a1 + 2 * edge_a1_bytes(R));

  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int e0 = tile * R;
  const int nvalid = min(R, p.E - e0);

  if (tid < R) {
    int d = 0, s = 0;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (tid < nvalid) {
      d = p.edge_dst[e0 + tid];
      s = p.edge_src[e0 + tid];
      dx = p.x[3 * d] - p.x[3 * s];
      dy = p.x[3 * d + 1] - p.x[3 * s + 1];
      dz = p.x[3 * d + 2] - p.x[3 * s + 2];
    }
    s_dst[tid] = d;
    s_src[tid] = s;
    s_diff[tid] = dx; s_diff[R + tid] = dy; s_diff[2 * R + tid] = dz;
    // reference: torch.norm(coords_i-coords_j,dim=1)**2 (sqrt then square, :56)
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    s_d2[tid] = nrm * nrm;
  }
  __syncthreads();

  // ---- coordinate branch ----
  switch (p.cbx) {
    case 2: phase_x<PREC, RB, 2>(p, s_dst, s_src, s_d2, s_a1, s_part); break;
    case 4: phase_x<PREC, RB, 4>(p, s_dst, s_src, s_d2, s_a1, s_part); break;
    default: phase_x<PREC, RB, 8>(p, s_dst, s_src, s_d2, s_a1, s_part); break;
  }
  __syncthreads();
  if (tid < R) s_sval[tid] = p.scal[0] + ((s_part[tid] + s_part[R + tid]) + (s_part[2 * R + tid] + s_part[3 * R + tid]));

  // ---- message branch ----
  switch (p.cbm) {
    case 2: phase_m<PREC, RB, 2>(p, s_dst, s_src, s_d2, s_a1, s_msg); break;
    case 4: phase_m<PREC, RB, 4>(p, s_dst, s_src, s_d2, s_a1, s_msg); break;
    default: phase_m<PREC, RB, 8>(p, s_dst, s_src, s_d2, s_a1, s_msg); break;
  }
  __syncthreads();

  // attention gate: sigmoid(wa . m + ba)  (:31-34, :59-60); kThreads/R threads per row
  {
    constexpr int TPR = (kThreads / R) >= 4 ? 4 : ((kThreads / R) >= 2 ? 2 : 1);
    const int row = tid / TPR, sub = tid % TPR;
    const int ld = p.MP + 1;
    float s = 0.f;
    if (row < R)
      for (int c = sub; c < p.MP; c += TPR) s = fmaf(p.wa[c], s_msg[row * ld + c], s);
    if (TPR >= 2) s += __shfl_xor(s, 1);
    if (TPR >= 4) s += __shfl_xor(s, 2);
    if (row < R && sub == 0) s_gate[row] = sigmoid_f(s + p.scal[1]);
  }
  __syncthreads();

  // ---- segment sums over the receiving node (aggr='sum' into edge_index[0], :10-11) ----
  // A node's edges are contiguous (CSR).  A segment that holds all edges of its node is stored
  // directly; otherwise it goes to the tile's partial slot (1 = holds the node's first edge,
  // 0 = continues a node started in an earlier tile) and node_post adds the partials in tile order.
  auto flush = [&](int n, int rs, int re, float v, float* agg, float* part, int ld, int c) {
    const bool first = (e0 + rs) == p.row_ptr[n];
    const bool last = (e0 + re + 1) == p.row_ptr[n + 1];
    if (first && last) agg[(size_t)n * ld + c] = v;
    else part[((size_t)tile * 2 + (first ? 1 : 0)) * ld + c] = v;
  };
  {
    const int ld = p.MP + 1;
    for (int c = tid; c < p.MP; c += kThreads) {
      float sum = 0.f;
      int rs = 0;
      for (int rr = 0; rr < nvalid; ++rr) {
        sum = fmaf(s_msg[rr * ld + c], s_gate[rr], sum);
        if (rr == nvalid - 1 || s_dst[rr + 1] != s_dst[rr]) {
          flush(s_dst[rr], rs, rr, sum, p.agg_m, p.part_m, p.MP, c);
          sum = 0.f;
          rs = rr + 1;
        }
      }
    }
    if (tid < 3) {  // coordinate messages (x_i - x_j) * s_ij; the 1/(G+1) factor is applied in node_post
      float sum = 0.f;
      int rs = 0;
      for (int rr = 0; rr < nvalid; ++rr) {
        sum = fmaf(s_diff[tid * R + rr], s_sval[rr], sum);
        if (rr == nvalid - 1 || s_dst[rr + 1] != s_dst[rr]) {
          flush(s_dst[rr], rs, rr, sum, p.agg_x, p.part_x, 4, tid);
          sum = 0.f;
          rs = rr + 1;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// node_post: h' = mlp_h([h | sum_m]) (:69), x' = x + sum_x / (G + 1) (:64, :70)
// ------------------------------------------------------------------------------------------------
constexpr int kPostNodes = 32;
constexpr int kPostHC = 512;   // hidden columns kept in LDS at a time
__host__ __device__ inline size_t post_smem_bytes(int K1P, int WhP) {
  const int hc = WhP < kPostHC ? WhP : kPostHC;
  size_t hs = (size_t)hc * 33 * 4, red = (size_t)4 * 16 * 64 * 4;
  return (size_t)K1P * 33 * 4 + (hs > red ? hs : red);
}

__global__ __launch_bounds__(kThreads, 1) void node_post_kernel(const PostParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Xs = reinterpret_cast<float*>(smem);  // [K1P][33]
  float* Hs = Xs + (size_t)p.K1P * 33;         // [HC][33], reused for the cross-wave reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n0 = blockIdx.x * kPostNodes;

  // gather [h | sum_m] for 32 nodes
  for (int i = tid; i < kPostNodes * p.K1P; i += kThreads) {
    const int node = i / p.K1P, k = i % p.K1P, n = n0 + node;
    float v = 0.f;
    if (n < p.N) {
      if (k < p.H) v = p.h[(size_t)n * p.H + k];
      else if (k - p.H < p.MP) {
        const int c = k - p.H, rp0 = p.row_ptr[n], rp1 = p.row_ptr[n + 1];
        if (rp1 > rp0) {
          const int t0 = rp0 / p.R, t1 = (rp1 - 1) / p.R;
          if (t0 == t1) v = p.agg_m[(size_t)n * p.MP + c];
          else {
            v = p.part_m[((size_t)t0 * 2 + 1) * p.MP + c];
            for (int t = t0 + 1; t <= t1; ++t) v += p.part_m[((size_t)t * 2) * p.MP + c];
          }
        }
      }
    }
    Xs[k * 33 + node] = v;
  }
  // coordinate update
  if (tid < kPostNodes * 3) {
    const int node = tid / 3, d = tid % 3, n = n0 + node;
    if (n < p.N) {
      const int rp0 = p.row_ptr[n], rp1 = p.row_ptr[n + 1];
      float v = 0.f;
      if (rp1 > rp0) {
        const int t0 = rp0 / p.R, t1 = (rp1 - 1) / p.R;
        for (int hs = 0; hs < p.nsplit_x; ++hs) {
          const float* ax = p.agg_x + (size_t)hs * p.agg_x_stride;
          const float* px = p.part_x + (size_t)hs * p.part_x_stride;
          if (t0 == t1) v += ax[(size_t)n * 4 + d];
          else {
            v += px[((size_t)t0 * 2 + 1) * 4 + d];
            for (int t = t0 + 1; t <= t1; ++t) v += px[((size_t)t * 2) * 4 + d];
          }
        }
      }
      const float g = 1.0f / (sqrtf(p.gscale[p.per_graph ? p.node_graph[n] : 0]) + 1.0f);
      p.x_out[3 * n + d] = p.x[3 * n + d] + v * g;
    }
  }
  __syncthreads();

  const int OB = p.HP / 32;
  const int KS4a = p.K1P / 8, KS4b = p.WhP / 8;
  f32x16 oacc[kPostMaxOB];
#pragma unroll
  for (int ob = 0; ob < kPostMaxOB; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[ob][i] = 0.f;

  for (int ch0 = 0; ch0 < p.WhP; ch0 += kPostHC) {
    const int hc = min(kPostHC, p.WhP - ch0);  // multiple of 128
    const int nbw = hc / 128;                  // column blocks per wave (1..4)
    // phase A: hidden chunk = SiLU(X . W1h^T + b1h)
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int ks4 = 0; ks4 < KS4a; ++ks4) {
      f32x4 b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nbw) b[j] = p.w1h[((size_t)(ch0 / 32 + wave * nbw + j) * KS4a + ks4) * 64 + lane];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float a = Xs[(8 * ks4 + 2 * s + hh) * 33 + r];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < nbw) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j][s], acc[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nbw) {
        const int cl = 32 * (wave * nbw + j) + r;  // column inside the chunk
        const float b = p.b1h[ch0 + cl];
#pragma unroll
        for (int i = 0; i < 16; ++i) Hs[cl * 33 + acc_row(i, lane)] = silu_f(acc[j][i] + b);
      }
    __syncthreads();
    // phase B: out += hidden chunk . W2h^T ; K split across the 4 waves
    const int kw = hc / 4;  // hidden columns per wave (multiple of 32)
    for (int q = 0; q < kw / 8; ++q) {
      const int kl = wave * kw + 8 * q;  // local hidden index of this group of 4 k-steps
      float a[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] = Hs[(kl + 2 * s + hh) * 33 + r];
#pragma unroll
      for (int ob = 0; ob < kPostMaxOB; ++ob)
        if (ob < OB) {
          const f32x4 b = p.w2h[((size_t)ob * KS4b + (ch0 + kl) / 8) * 64 + lane];
#pragma unroll
          for (int s = 0; s < 4; ++s) oacc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], oacc[ob], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  // cross-wave reduction + store
  float* red = Hs;  // [4][16][64]
#pragma unroll
  for (int ob = 0; ob < kPostMaxOB; ++ob) {
    if (ob < OB) {
#pragma unroll
      for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = oacc[ob][i];
      __syncthreads();
      for (int e = tid; e < 16 * 64; e += kThreads) {
        const int i = e >> 6, l = e & 63;
        const float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
        const int node = acc_row(i, l), col = 32 * ob + (l & 31), n = n0 + node;
        if (n < p.N && col < p.H) p.h_out[(size_t)n * p.H + col] = v + p.b2h[col];
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int edge_rows_per_tile(int prec) { (void)prec; return 64; }

template <typename T>
static int dev_alloc(T** p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (count == 0) return EGNN_OK;
  if (hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)) != hipSuccess) {
    set_error("hipMalloc of %zu bytes failed", count * sizeof(T));
    *p = nullptr;
    return EGNN_ENOMEM;
  }
  // EGNN_DEBUG_POISON=1 (tests): every scratch / pack buffer starts as 0xFF bytes -- NaN as fp32, fp16 and bf16, -1 as an index --
  // so that an element a kernel reads before any kernel wrote it cannot pass for a plausible value (hipMalloc hands out zeros
  // in a fresh process and whatever the previous owner left afterwards: tests/test_gpu_parity.py::test_no_scratch_read_before_write)
  const char* poison = getenv("EGNN_DEBUG_POISON");
  if (poison && poison[0] == '1' && hipMemset(*p, 0xFF, count * sizeof(T)) != hipSuccess) {
    set_error("hipMemset (EGNN_DEBUG_POISON) failed");
    return EGNN_EHIP;
  }
  return EGNN_OK;
}

int reserve(egnn_ctx* c) {
  if (c->L == 0 || c->N == 0) { set_error("model and graph must be set first"); return EGNN_ESTATE; }
  const int R = edge_rows_per_tile(0);
  // partial slots per tile: 64-row tiles are the smallest, except for small graphs, which may run on 32-row tiles
  c->small_ok = c->E <= (1 << 16);
  const size_t tiles = c->small_ok ? (size_t)(c->E + 31) / 32 : (size_t)(c->E + R - 1) / R;
  int rc = EGNN_OK;
  if ((size_t)c->N > c->cap_nodes) {
    const size_t n = c->N;
    if ((rc = dev_alloc(&c->table, n * c->TC))) return rc;
    if ((rc = dev_alloc(&c->agg_m, n * c->MP))) return rc;
    if ((rc = dev_alloc(&c->agg_x, 4 * n * 4))) return rc;   // up to 4 column-split copies
    if ((rc = dev_alloc(&c->node_d2, n))) return rc;
    if ((rc = dev_alloc(&c->h_partial, n <= 1024 ? (size_t)8 * n * c->H : 0))) return rc;
    for (int i = 0; i < 2; ++i) {
      if ((rc = dev_alloc(&c->h_tmp[i], n * c->H))) return rc;
      if ((rc = dev_alloc(&c->x_tmp[i], n * 3))) return rc;
    }
    c->cap_nodes = n;
  }
  if (tiles > c->cap_tiles) {
    if ((rc = dev_alloc(&c->part_m, (tiles + 1) * 2 * c->MP))) return rc;
    if ((rc = dev_alloc(&c->part_x, 4 * (tiles + 1) * 2 * 4))) return rc;
    c->cap_tiles = tiles;
  }
  if (!c->stamps) {
    if ((rc = dev_alloc(&c->stamps, (size_t)2 * 8 * 32 * 4))) return rc;
    EGNN_HIP(hipMemset(c->stamps, 0, 2 * 8 * 32 * 4 * 8));
  }
  if ((size_t)c->B > c->cap_graphs) {
    if ((rc = dev_alloc(&c->gscale, (size_t)c->B + 1))) return rc;
    c->cap_graphs = c->B;
  }
  return EGNN_OK;
}

static void prof_begin(egnn_ctx* c, hipStream_t st, int kind) {
  if (!c->prof) return;
  if (c->ev_used + 2 > c->ev.size()) {
    for (int i = 0; i < 2; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      c->ev.push_back(e);
    }
    c->ev_kind.push_back(kind);
  } else {
    c->ev_kind[c->ev_used / 2] = kind;
  }
  (void)hipEventRecord(c->ev[c->ev_used], st);
}
static void prof_end(egnn_ctx* c, hipStream_t st) {
  if (!c->prof || c->ev_used + 2 > c->ev.size()) return;
  (void)hipEventRecord(c->ev[c->ev_used + 1], st);
  c->ev_used += 2;
}

// Side stream + fork / join events of a context.  The library creates NO stream of its own: the caller hands one in
// (egnn_set_side_stream; the Python host passes the process-wide auxiliary stream of diffusion_model_amd/streams.py, the same
// one the gradient all-reduce uses), so a process holds a fixed, small set of HIP streams however many contexts and samplers
// it creates.  Round 2 created a stream per sampler context and saw, in the 2-processes-on-one-GPU gloo rehearsal, every
// gradient all-reduce take seconds once one more stream existed: two processes with 5-6 streams each oversubscribe the
// device's hardware queues, and the scheduler then multiplexes queues between the processes with a coarse time slice
// (DESIGN.md section 7).
// The coordinate and the message kernel of a layer are independent.  On one stream the message kernel starts when the LAST
// round of coordinate workgroups has drained; when that round leaves enough CUs idle for all message workgroups (two per
// CU), the message kernel goes to the side stream and runs in that shadow: one 64-atom graph 64 of 256 CUs busy, five graphs
// (generate()'s default gen_num_per_spectrum) 316 coordinate workgroups = 1.23 rounds -> 0.49 -> 0.42 ms per reverse step.
// Full rounds (16 graphs and more: measured equal eager, 2-8 % slower in graph replay) keep the single stream.
// (the decision itself: fork_candidate() in host_logic.cpp)

int fork_streams(egnn_ctx* c) {   // fork / join events for a context whose caller provided a side stream
  if (!c->side || c->ev_fork) return EGNN_OK;
  if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) return EGNN_EHIP;
  return EGNN_OK;
}

int init_kernel_attributes() {
  // raise the dynamic-LDS limit of every big kernel once, outside any stream capture
  static bool done = false;
  if (done) return EGNN_OK;
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel<EGNN_PREC_BF16, 2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel<EGNN_PREC_F32, 2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&node_post_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&node_pre_hilo_kernel<8>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&node_pre_hilo_kernel<2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int rc = init_edge_bf16_v3_attributes();
  if (rc) return rc;
  if ((rc = init_edge_bf16_v4_attributes())) return rc;
  if ((rc = init_node_bf16_attributes())) return rc;
  if ((rc = init_edge_dgrad_attributes())) return rc;
  if ((rc = init_edge_dgrad_graph_attributes())) return rc;
  if ((rc = init_edge_x_m16_attributes())) return rc;
  if ((rc = init_edge_bf16x3_attributes())) return rc;
  if ((rc = init_edge_f16c8_attributes())) return rc;
  if ((rc = init_edge_f16c8w_attributes())) return rc;
  done = true;
  return EGNN_OK;
}

template <int PREC, int RB>
static int launch_edge(const EdgeParams& p, int tiles, size_t smem, hipStream_t st) {
  hipLaunchKernelGGL((edge_kernel<PREC, RB>), dim3(tiles), dim3(kThreads), smem, st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

// 32-edge tiles (edge_small.hip) for graphs of up to ~2 k edges (a 20-40-atom molecule, the toy graphs of configs[0]): measured
// (profiles/r04e_latency_small_tiles.log, hipGraph replay, ms per reverse step) 20-atom cell 0.289 -> 0.277, 4 x 2-atom graphs
// 0.258 -> 0.223; but one 64-atom cell (4,032 edges) 0.313 -> 0.343 and five 0.41 -> 0.70: a layer's weight traffic through L2
// is (E / tile rows) x 2.5 MB whatever the column split, 315 MB at 32 rows for ONE 64-atom graph, and the chip's L2 delivers
// ~7 TB/s to 256 streaming CUs, not the 34 TB/s of 64 B/clk/CU -- 128-row tiles are within 3x of that floor already.
// EGNN_SMALL_EDGES overrides the edge-count limit (0 = never): measurement switch.
// -> edges per tile (32 / 64) or 0 = the 128-edge-tile kernels.  EGNN_SMALL64_EDGES: limit of the 64-row form (one 64-atom graph).
static int small_tiles(const egnn_ctx* c, const EdgeParams& p) {
  static const long limit = getenv("EGNN_SMALL_EDGES") ? atol(getenv("EGNN_SMALL_EDGES")) : 2048;
  static const long limit64 = getenv("EGNN_SMALL64_EDGES") ? atol(getenv("EGNN_SMALL64_EDGES")) : 6144;
  if (c->E <= 0 || !c->small_ok || !edge_small_supported(p)) return 0;
  return (long)c->E <= limit ? 32 : ((long)c->E <= limit64 ? 64 : 0);
}

// EdgeParams of layer `layer` over the graph set on the context (unscaled parameter vectors, fp32 / bf16 fragments)
static void fill_edge_params(egnn_ctx* c, int layer, int prec, const float* x, EdgeParams& p) {
  const LayerPack& lp = c->layers[layer];
  memset(&p, 0, sizeof(p));
  p.N = c->N; p.E = c->E;
  p.edge_dst = c->edge_dst; p.edge_src = c->edge_src; p.row_ptr = c->row_ptr;
  p.x = x; p.table = c->table;
  p.TC = c->TC; p.WxP = c->WxP; p.WmP = c->WmP; p.MP = c->MP; p.cbx = c->cbx; p.cbm = c->cbm;
  p.wdx = lp.wdx; p.wdm = lp.wdm; p.b2x = lp.b2x; p.w3x = lp.w3x; p.b2m = lp.b2m; p.wa = lp.wa;
  p.scal = lp.scal;
  p.w2x = prec == EGNN_PREC_BF16 ? lp.w2x_bf16 : (const void*)lp.w2x_f32;
  p.w2m = prec == EGNN_PREC_BF16 ? lp.w2m_bf16 : (const void*)lp.w2m_f32;
  p.agg_m = c->agg_m; p.agg_x = c->agg_x; p.part_m = c->part_m; p.part_x = c->part_x;
  p.agg_x_stride = (size_t)c->cap_nodes * 4; p.part_x_stride = (c->cap_tiles + 1) * 2 * 4;
  p.stamps = c->stamps;
#ifdef EGNN_DIAG   // diagnostic builds only (diag.h): EGNN_DEBUG switches the weight / table streams off
  static const int dbg = getenv("EGNN_DEBUG") ? atoi(getenv("EGNN_DEBUG")) : 0;
  p.dbg = dbg;
#endif
}
// the bf16 fast kernels (v2..v4) use the copies pre-scaled by -log2(e) / -1/log2(e) (LayerPack::sc)
static void use_scaled_pack(egnn_ctx* c, int layer, EdgeParams& p, const float*& w1catT, const float*& b1cat) {
  const LayerPack& lp = c->layers[layer];
  const float* o = lp.sc;
  w1catT = o; o += (size_t)c->H * c->TC;
  b1cat = o; o += c->TC;
  p.wdx = o; o += c->WxP;
  p.wdm = o; o += c->WmP;
  p.b2x = o; o += c->WxP;
  p.w3x = o; o += c->WxP;
  p.b2m = o; o += c->MP;
  p.wa = o;
  p.w2x = lp.w2x_bf16s; p.w2m = lp.w2m_bf16s;
  p.w2x16 = lp.w2x_bf16s16; p.w2m16 = lp.w2m_bf16s16;
  p.w2x_lo = lp.w2x_bf16s_lo; p.w2m_lo = lp.w2m_bf16s_lo;
  p.w2x_c8 = lp.w2x_c8; p.w2m_c8 = lp.w2m_c8; p.c8_exp = lp.c8_exp;
}
// first-layer table of the v3 / v4 kernels (fp16, pre-scaled) for node features h
static int launch_node_pre_f16(egnn_ctx* c, hipStream_t st, int layer, const float* h, const float* w1catT,
                               const float* b1cat) {
  const int N = c->N;
  if (c->H <= 48) {   // bf16 hi/lo MFMA (2^-16), else exact f32 MFMA, else the VALU kernel
    const int ntile = (N + kPre3Nodes - 1) / kPre3Nodes;
    const size_t sm = (size_t)48 * 33 * 4 + (size_t)4 * 32 * (128 + 8) * 2;
    const bf16x8* w1hl = reinterpret_cast<const bf16x8*>(c->layers[layer].w1hl_bf16);
    _Float16* tab = reinterpret_cast<_Float16*>(c->table);
    // a pending hidden-split finish of the previous layer (c->pend) rides in this launch
    const bool fin = c->pend.active && c->pend.h_out == h;
    const int fhs = fin ? c->pend.hs : 0;
    const float* fb = fin ? c->pend.b2h : nullptr;
    float* fw = fin ? c->pend.h_out : nullptr;
    if (fin) c->pend.active = false;
    if (ntile * (c->TC / 1024) >= 256)
      hipLaunchKernelGGL(node_pre_hilo_kernel<8>, dim3(ntile, (c->TC + 1023) / 1024), dim3(kThreads), sm, st, h, N, c->H, w1hl,
                         b1cat, c->TC, tab, fhs, c->h_partial, fb, fw);
    else
      hipLaunchKernelGGL(node_pre_hilo_kernel<2>, dim3(ntile, (c->TC + 255) / 256), dim3(kThreads), sm, st, h, N, c->H, w1hl,
                         b1cat, c->TC, tab, fhs, c->h_partial, fb, fw);
  } else if (c->H <= 64) {
    dim3 grid((N + kPre2Nodes - 1) / kPre2Nodes, (c->TC + kPre2Cols - 1) / kPre2Cols);
    const size_t sm = (size_t)((c->H + 1) & ~1) * 33 * sizeof(float);
    const size_t sm16 = (size_t)kPre2Nodes * (kPre2Cols + 8) * 2;   // output staging tile of the fp16 variant
    hipLaunchKernelGGL(node_pre_mfma_kernel<_Float16>, grid, dim3(kThreads), sm > sm16 ? sm : sm16, st, h, N, c->H,
                       w1catT, b1cat, c->TC, reinterpret_cast<_Float16*>(c->table));
  } else {
    dim3 grid((N + kPreNodes - 1) / kPreNodes, (c->TC + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(node_pre_kernel<_Float16>, grid, dim3(kThreads), (size_t)kPreNodes * c->H * sizeof(float), st, h, N,
                       c->H, w1catT, b1cat, c->TC, reinterpret_cast<_Float16*>(c->table));
  }
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

// Backward recompute (bf16 fast path), see egcl_backward_edge_recompute in include/egnn_amd.h
int backward_recompute_supported(egnn_ctx* c) {
  EdgeParams p;
  fill_edge_params(c, 0, EGNN_PREC_BF16, nullptr, p);
  return edge_bf16_v4_supported(p) && edge_bf16_v3_supported(p) && c->Wx == c->WxP && c->Wm == c->WmP && c->M == c->MP;
}
int backward_table(egnn_ctx* c, hipStream_t st, int layer, const float* h) {
  EdgeParams p;
  fill_edge_params(c, layer, EGNN_PREC_BF16, nullptr, p);
  const float *w1catT, *b1cat;
  use_scaled_pack(c, layer, p, w1catT, b1cat);
  return launch_node_pre_f16(c, st, layer, h, w1catT, b1cat);
}
int backward_recompute(egnn_ctx* c, hipStream_t st, int layer, const float* x, const float* g_sum_x, const float* g_sum_m,
                       int e_first, int n_edges, void* s1x, void* s1m, void* g_a2x, void* g_a2m, float* s_halves,
                       float* g_b2x, float* g_w3, float* g_b3, float* g_b2m, float* g_wa, float* g_ba) {
  EdgeParams p;
  fill_edge_params(c, layer, EGNN_PREC_BF16, x, p);
  const float *w1catT, *b1cat;
  use_scaled_pack(c, layer, p, w1catT, b1cat);
  p.edge_dst = c->edge_dst + e_first; p.edge_src = c->edge_src + e_first; p.E = n_edges;
  p.g_sum_x = g_sum_x; p.g_sum_m = g_sum_m;
  p.s1_out = s1x; p.g_a2_out = g_a2x; p.s_half_out = s_halves;
  p.g_col_a = g_b2x; p.g_col_b = g_w3; p.g_scalar = g_b3;
  int rc = launch_edge_bf16_v3_x_bwd(p, st);
  if (rc) return rc;
  p.s1_out = s1m; p.g_a2_out = g_a2m; p.s_half_out = nullptr;
  p.g_col_a = g_b2m; p.g_col_b = g_wa; p.g_scalar = g_ba;
  return launch_edge_bf16_v4_m_bwd(p, st);
}

// dL/da2 of both edge MLPs from the pre-activations egcl_forward_save left (in place), see edge_bwd_heads.hip
int backward_heads_saved(egnn_ctx* c, hipStream_t st, int layer, const float* x, const float* g_sum_x, const float* g_sum_m,
                         int e_first, int n_edges, void* t2x, void* t2m, float* g_b2x, float* g_w3, float* g_b3, float* g_b2m,
                         float* g_wa, float* g_ba) {
  EdgeParams p;
  fill_edge_params(c, layer, EGNN_PREC_BF16, x, p);
  const float *w1catT, *b1cat;
  use_scaled_pack(c, layer, p, w1catT, b1cat);
  return launch_heads_saved(n_edges, c->edge_dst + e_first, c->edge_src + e_first, x, g_sum_x, g_sum_m, c->WxP, c->MP, p.w3x, p.wa,
                            p.scal, t2x, t2m, g_b2x, g_w3, g_b3, g_b2m, g_wa, g_ba, st);
}

int backward_dgrad(egnn_ctx* c, hipStream_t st, int layer, const float* x, int e_first, int n_edges, const void* g_a2x,
                   const void* g_a2m, void* g_a1x, void* g_a1m) {
  const LayerPack& lp = c->layers[layer];
  const float* wdx_s = lp.sc + (size_t)(c->H + 1) * c->TC;   // scaled copies: [w1catT | b1cat | wdx | wdm | ...]
  const float* wdm_s = wdx_s + c->WxP;
  int rc = launch_edge_dgrad(c->N, n_edges, c->edge_dst + e_first, c->edge_src + e_first, x, c->table, c->TC, 0, c->WxP, wdx_s,
                             g_a2x, c->WxP, lp.w2xT_bf16, c->WxP, g_a1x, st);
  if (rc) return rc;
  return launch_edge_dgrad(c->N, n_edges, c->edge_dst + e_first, c->edge_src + e_first, x, c->table, c->TC, 2 * c->WxP,
                           2 * c->WxP + c->WmP, wdm_s, g_a2m, c->MP, lp.w2mT_bf16, c->WmP, g_a1m, st);
}

// the same dgrad WITHOUT dL/da1 in memory: per-graph workgroups that also reduce it for the first Linear layers
// (edge_bwd_dgrad_graph.hip); G = bf16 [N][2 WxP + 2 WmP] = [Gd_x | Gs_x | Gd_m | Gs_m], gd2_part = [(WxP + WmP) / 256][n_edges]
int backward_dgrad_graph(egnn_ctx* c, hipStream_t st, int layer, const float* x, int e_first, int n_edges, const void* g_a2x,
                         const void* g_a2m, void* G, float* cd_x, float* cd_m, float* gd2_part) {
  const LayerPack& lp = c->layers[layer];
  const float* wdx_s = lp.sc + (size_t)(c->H + 1) * c->TC;
  const float* wdm_s = wdx_s + c->WxP;
  const int ldg = 2 * c->WxP + 2 * c->WmP;
  __bf16* g = static_cast<__bf16*>(G);
  int rc = launch_edge_dgrad_graph(c->N, c->B, c->graph_ptr, c->row_ptr, c->edge_dst, c->edge_src, e_first, n_edges, x, c->table,
                                   c->TC, 0, c->WxP, wdx_s, g_a2x, c->WxP, lp.w2xT_bf16, c->WxP, g, g + c->WxP, ldg, cd_x,
                                   gd2_part, st);
  if (rc) return rc;
  return launch_edge_dgrad_graph(c->N, c->B, c->graph_ptr, c->row_ptr, c->edge_dst, c->edge_src, e_first, n_edges, x, c->table,
                                 c->TC, 2 * c->WxP, 2 * c->WxP + c->WmP, wdm_s, g_a2m, c->MP, lp.w2mT_bf16, c->WmP,
                                 g + 2 * c->WxP, g + 2 * c->WxP + c->WmP, ldg, cd_m,
                                 gd2_part + (size_t)(c->WxP / 256) * n_edges, st);
}

// Stage 1 of a layer: node_pre, squared-distance sums and the fused edge pass.  gsum (c->gscale) then holds the
// sum of d^2 over the edges THIS context received, per graph (or per call).
int launch_layer_begin(egnn_ctx* c, hipStream_t st, int layer, int prec, int norm_scope, const float* h,
                       const float* x, bool need_gscale) {
  const LayerPack& lp = c->layers[layer];
  if (!lp.packed) { set_error("layer %d has no packed parameters", layer); return EGNN_ESTATE; }
  const int N = c->N, E = c->E;
  int R = edge_rows_per_tile(prec), nsplit_x = 1;
  const int per_graph = norm_scope == EGNN_NORM_GRAPH;

  // choose the edge path first: the 128-edge-tile bf16 kernels consume the pre-scaled first-layer table
  //   path 5  precision bf16x3: edge_bf16x3.hip (head / remainder operands, fp32 table)
  //   path 4  precision bf16  : coordinate kernel edge_x_m16.hip (hidden width 512 / 1024) or edge_bf16_v3.hip (256) +
  //                             message kernel edge_bf16_v4.hip (fp16 table)
  //   path 1  every other shape, and fp32: the generic 64-edge-tile kernel edge_kernel<PREC> of this file
  // EGNN_EDGE=1 forces path 1 (the one switch kept: the fallback kernels' own parity test runs the reference widths on them)
  EdgeParams p;
  fill_edge_params(c, layer, prec, x, p);
  static const int edge_sel = getenv("EGNN_EDGE") ? atoi(getenv("EGNN_EDGE")) : 4;
  int path = 1;
  if (prec == EGNN_PREC_BF16X3) {   // split-operand kernels where the 128-edge tiling applies, else the exact fp32 path
    EdgeParams q = p;
    const float *w1c, *b1c;
    use_scaled_pack(c, layer, q, w1c, b1c);
    if (edge_sel >= 4 && edge_bf16x3_supported(q)) path = 5;
    else prec = EGNN_PREC_F32;
    fill_edge_params(c, layer, prec, x, p);
  }
  if (prec == EGNN_PREC_F16C8) {   // fp16 heads + e4m3 corrections where the 128-edge tiling applies, else the exact fp32 path
    EdgeParams q = p;
    const float *w1c, *b1c;
    use_scaled_pack(c, layer, q, w1c, b1c);
    q.w2x16 = lp.w2x_f16s16; q.w2m16 = lp.w2m_f16s16;
    if (edge_sel >= 4 && edge_f16c8_supported(q) && !c->save_s1x) path = 7;
    else prec = EGNN_PREC_F32;
    fill_edge_params(c, layer, prec, x, p);
  }
  if (prec == EGNN_PREC_F16) {   // fp16 operands on the bf16 path's kernels (hidden width 512 / 1024), else the exact fp32 path
    EdgeParams q = p;
    const float *w1c, *b1c;
    use_scaled_pack(c, layer, q, w1c, b1c);
    if (edge_sel >= 4 && edge_bf16_v4_supported(q) && edge_x_m16_supported(q) && !c->save_s1x) path = 6;
    else { prec = EGNN_PREC_F32; fill_edge_params(c, layer, prec, x, p); }
  }
  if (prec == EGNN_PREC_BF16 && edge_sel >= 4 && edge_bf16_v4_supported(p) && edge_bf16_v3_supported(p)) path = 4;
  if (c->save_s1x && path != 4) { set_error("egcl_forward_save needs the 128-edge-tile bf16 kernels"); return EGNN_EINVAL; }
  const float* w1catT = lp.w1catT;
  const float* b1cat = lp.b1cat;
  if (path >= 4) use_scaled_pack(c, layer, p, w1catT, b1cat);

  // a pending hidden-split finish (previous layer of a multi-layer call) is fused into the half-precision node_pre for H <= 48;
  // every other case runs the finish launch now
  if (c->pend.active && !((path == 4 || path == 6) && c->H <= 48 && c->pend.h_out == h)) {
    int rc = launch_node_post_finish(N, c->H, c->pend.hs, c->h_partial, c->pend.b2h, c->pend.h_out, st);
    c->pend.active = false;
    if (rc) return rc;
  }
  prof_begin(c, st, 1);
  {
    if (path == 5 || path == 7) {   // bf16x3 / f16c8: exact fp32 table of the scaled first layers
      dim3 grid((N + kPre2Nodes - 1) / kPre2Nodes, (c->TC + kPre2Cols - 1) / kPre2Cols);
      const size_t sm = (size_t)((c->H + 1) & ~1) * 33 * sizeof(float);
      if (c->H <= 64)
        hipLaunchKernelGGL(node_pre_mfma_kernel<float>, grid, dim3(kThreads), sm, st, h, N, c->H, w1catT, b1cat, c->TC, c->table);
      else
        hipLaunchKernelGGL(node_pre_kernel<float>, dim3((N + kPreNodes - 1) / kPreNodes, (c->TC + kThreads - 1) / kThreads),
                           dim3(kThreads), (size_t)kPreNodes * c->H * sizeof(float), st, h, N, c->H, w1catT, b1cat, c->TC, c->table);
    } else if (path == 4 || path == 6) {   // half-precision table
      int rc = launch_node_pre_f16(c, st, layer, h, w1catT, b1cat);
      if (rc) return rc;
    } else if (c->H <= 64) {
      dim3 grid((N + kPre2Nodes - 1) / kPre2Nodes, (c->TC + kPre2Cols - 1) / kPre2Cols);
      const size_t sm = (size_t)((c->H + 1) & ~1) * 33 * sizeof(float);
      hipLaunchKernelGGL(node_pre_mfma_kernel<float>, grid, dim3(kThreads), sm, st, h, N, c->H, w1catT, b1cat, c->TC,
                         c->table);
    } else {
      dim3 grid((N + kPreNodes - 1) / kPreNodes, (c->TC + kThreads - 1) / kThreads);
      const size_t sm = (size_t)kPreNodes * c->H * sizeof(float);
      hipLaunchKernelGGL(node_pre_kernel<float>, grid, dim3(kThreads), sm, st, h, N, c->H, w1catT, b1cat, c->TC,
                         c->table);
    }
    if (path < 4) {   // the 128-edge-tile kernels sum d^2 per receiving node themselves
      hipLaunchKernelGGL(node_d2_kernel, dim3((N + 255) / 256), dim3(256), 0, st, x, c->row_ptr, c->edge_src, N,
                         c->node_d2);
      hipLaunchKernelGGL(graph_sum_kernel, dim3(per_graph ? c->B : 1), dim3(kThreads), 0, st, c->node_d2,
                         c->graph_ptr, N, per_graph, c->gscale);
    }
  }
  prof_end(c, st);
  EGNN_HIP(hipGetLastError());

  if (E > 0) {
    const int tiles = (E + R - 1) / R;
    const size_t smem = edge_smem_bytes(R, c->MP);
    prof_begin(c, st, 0);
    int rc;
    if (path == 5) {
      R = 128;
      nsplit_x = p.WxP / 256;
      rc = launch_edge_bf16x3(p, st);
    } else if (path == 7) {   // precision f16c8 (edge_f16c8.hip): fp16 16-column streams + e4m3 correction streams
      R = 128;
      p.w2x16 = lp.w2x_f16s16; p.w2m16 = lp.w2m_f16s16;
      // matrix tile shape: 32x32 (edge_f16c8w.hip) / 16x16 (edge_f16c8.hip); EGNN_C8_TILE = A/B switch
      static const int c8_tile = getenv("EGNN_C8_TILE") ? atoi(getenv("EGNN_C8_TILE")) : 32;
      bool wide = false;
      if (c8_tile == 32) {
        EdgeParams q = p;
        q.w2x = lp.w2x_f16s; q.w2m = lp.w2m_f16s; q.w2x_c8 = lp.w2x_c8w; q.w2m_c8 = lp.w2m_c8w;
        if (edge_f16c8w_supported(q)) { p = q; wide = true; }
      }
      nsplit_x = wide ? p.WxP / 512 : edge_f16c8_x_split(p.WxP);
      const bool fork = !c->prof && st != nullptr && c->side != nullptr && c->ev_fork != nullptr && fork_candidate(E, p.WxP);
      if (fork) {
        EGNN_HIP(hipEventRecord(c->ev_fork, st));
        EGNN_HIP(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        rc = wide ? launch_edge_f16c8w_x(p, st) : launch_edge_f16c8_x(p, st);
        if (!rc) rc = wide ? launch_edge_f16c8w_m(p, c->side) : launch_edge_f16c8_m(p, c->side);
        EGNN_HIP(hipEventRecord(c->ev_join, c->side));
        EGNN_HIP(hipStreamWaitEvent(st, c->ev_join, 0));
      } else {
        rc = wide ? launch_edge_f16c8w_x(p, st) : launch_edge_f16c8_x(p, st);
        if (!rc) rc = wide ? launch_edge_f16c8w_m(p, st) : launch_edge_f16c8_m(p, st);
      }
    } else if ((path == 6 || path == 4) && !c->save_s1x && small_tiles(c, p) != 0) {
      // small graphs (the reference's per-call workload): 32-edge tiles, weight-stream-bound workgroups (edge_small.hip);
      // coordinate and message kernel side by side when the caller gave a side stream
      const bool f16 = path == 6;
      R = small_tiles(c, p);
      nsplit_x = p.WxP / 512;
      if (f16) { p.w2x16 = lp.w2x_f16s16; p.w2m16 = lp.w2m_f16s16; }
      const bool fork = !c->prof && st != nullptr && c->side != nullptr && c->ev_fork != nullptr;
      if (fork) {
        EGNN_HIP(hipEventRecord(c->ev_fork, st));
        EGNN_HIP(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        rc = launch_edge_small_x(p, st, f16, R);
        if (!rc) rc = launch_edge_small_m(p, c->side, f16, R);
        EGNN_HIP(hipEventRecord(c->ev_join, c->side));
        EGNN_HIP(hipStreamWaitEvent(st, c->ev_join, 0));
      } else {
        rc = launch_edge_small_x(p, st, f16, R);
        if (!rc) rc = launch_edge_small_m(p, st, f16, R);
      }
    } else if (path == 6) {   // precision fp16: the path-4 kernels on fp16 operands (same tiles, same launch structure)
      R = edge_v4_rows();
      nsplit_x = p.WxP / 512;
      p.w2x16 = lp.w2x_f16s16; p.w2m = lp.w2m_f16s;
      const bool fork = !c->prof && st != nullptr && c->side != nullptr && c->ev_fork != nullptr && fork_candidate(E, p.WxP);
      if (fork) {
        EGNN_HIP(hipEventRecord(c->ev_fork, st));
        EGNN_HIP(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        rc = launch_edge_x_m16_f16(p, st);
        if (!rc) rc = launch_edge_f16_v4_m(p, c->side);
        EGNN_HIP(hipEventRecord(c->ev_join, c->side));
        EGNN_HIP(hipStreamWaitEvent(st, c->ev_join, 0));
      } else {
        rc = launch_edge_x_m16_f16(p, st);
        if (!rc) rc = launch_edge_f16_v4_m(p, st);
      }
    } else if (path == 4) {
      R = edge_v4_rows();
      const bool xm16 = edge_x_m16_supported(p);   // hidden width 512 / 1024: v_mfma_f32_16x16x32_bf16 (512 columns per workgroup)
      nsplit_x = xm16 ? p.WxP / 512 : 1;
      // fork_candidate(): the message kernel goes to the caller's side stream between two events (fork / join; under capture
      // they become graph edges) and runs in the shadow of the coordinate kernel's last, partly filled round
      const bool fork = !c->prof && st != nullptr && c->side != nullptr && c->ev_fork != nullptr && fork_candidate(E, p.WxP);
      auto launch_x = [&](hipStream_t s) { return xm16 ? launch_edge_x_m16(p, s) : launch_edge_bf16_v3_x(p, s); };
      if (c->save_s1x) {   // training forward (egcl_forward_save): the same kernels, which also store what the backward needs
        p.s1_out = c->save_s1x; p.g_a2_out = c->save_t2x; p.s_half_out = c->save_s;
        rc = xm16 ? launch_edge_x_m16_save(p, st) : launch_edge_bf16_v3_x_save(p, st);
        p.s1_out = c->save_s1m; p.g_a2_out = c->save_t2m; p.s_half_out = nullptr;
        if (!rc) rc = launch_edge_bf16_v4_m_save(p, st);
      } else if (fork) {
        EGNN_HIP(hipEventRecord(c->ev_fork, st));
        EGNN_HIP(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        rc = launch_x(st);
        if (!rc) rc = launch_edge_bf16_v4_m(p, c->side);
        EGNN_HIP(hipEventRecord(c->ev_join, c->side));
        EGNN_HIP(hipStreamWaitEvent(st, c->ev_join, 0));
      } else {
        rc = launch_x(st);
        if (!rc) rc = launch_edge_bf16_v4_m(p, st);
      }
    } else if (prec == EGNN_PREC_BF16) rc = launch_edge<EGNN_PREC_BF16, 2>(p, tiles, smem, st);
    else rc = launch_edge<EGNN_PREC_F32, 2>(p, tiles, smem, st);
    prof_end(c, st);
    if (rc) return rc;
  }

  c->last_R = R; c->last_nsplit_x = nsplit_x; c->last_path = path;
  c->sq_from_agg = false;
  if (path >= 4) {
    if (E == 0) {
      EGNN_HIP(hipMemsetAsync(c->gscale, 0, sizeof(float) * (per_graph ? c->B : 1), st));
    } else if (!per_graph || need_gscale) {
      hipLaunchKernelGGL(graph_sq_sums_kernel, dim3(per_graph ? c->B : 1), dim3(kThreads), 0, st, c->row_ptr, R, c->agg_x,
                         c->part_x, c->graph_ptr, N, per_graph, c->gscale);
      EGNN_HIP(hipGetLastError());
    } else {
      c->sq_from_agg = true;   // node_post adds the per-graph sums up itself: no launch
    }
  }
  return EGNN_OK;
}

// Stage 2: node_post with the normaliser sums in c->gscale (possibly replaced by a cross-rank total).
int launch_layer_end(egnn_ctx* c, hipStream_t st, int layer, int prec, int norm_scope, const float* h, const float* x,
                     float* h_out, float* x_out, bool defer_finish = false) {
  const LayerPack& lp = c->layers[layer];
  const int N = c->N, R = c->last_R, nsplit_x = c->last_nsplit_x;
  const size_t agg_x_stride = (size_t)c->cap_nodes * 4, part_x_stride = (c->cap_tiles + 1) * 2 * 4;
  const int per_graph = norm_scope == EGNN_NORM_GRAPH;
  (void)prec;
  {
    PostParams q;
    q.N = N; q.H = c->H; q.MP = c->MP; q.K1P = c->K1P; q.WhP = c->WhP; q.HP = c->HP; q.R = R;
    q.h = h; q.x = x; q.row_ptr = c->row_ptr; q.node_graph = c->node_graph;
    q.agg_m = c->agg_m; q.agg_x = c->agg_x; q.part_m = c->part_m; q.part_x = c->part_x;
    q.gscale = c->gscale; q.per_graph = per_graph;
    q.graph_ptr = c->graph_ptr; q.sq_from_agg = c->sq_from_agg ? 1 : 0;
    q.agg_x_stride = agg_x_stride; q.part_x_stride = part_x_stride; q.nsplit_x = nsplit_x;
    q.w1h = reinterpret_cast<const f32x4*>(lp.w1h_f32); q.w2h = reinterpret_cast<const f32x4*>(lp.w2h_f32);
    q.b1h = lp.b1h; q.b2h = lp.b2h; q.h_out = h_out; q.x_out = x_out;
    q.w1h_bf16 = lp.w1h_bf16; q.w2h_bf16p = lp.w2h_bf16p; q.K1Q = c->K1Q;
    q.h_partial = c->h_partial;
    prof_begin(c, st, 1);
    // node MLP of precision fp16 (and of bf16x3 on the 128-edge-tile path): split-operand fp16 products (22 significant bits,
    // node_bf16.hip) where the shape allows, else the exact fp32 kernel.  EGNN_F16_NODE (error-budget experiments only):
    // 0 = exact fp32 node MLP, 1 = plain fp16 operands.
    static const int f16_node = getenv("EGNN_F16_NODE") ? atoi(getenv("EGNN_F16_NODE")) : 2;
    PostParams qs = q;
    qs.w1h_bf16 = lp.w1h_f16k; qs.w2h_bf16p = lp.w2h_f16p; qs.w1h_lo = lp.w1h_f16k_lo; qs.w2h_lo = lp.w2h_f16p_lo;
    const bool half_path = (prec == EGNN_PREC_F16 && c->last_path == 6) || (prec == EGNN_PREC_BF16X3 && c->last_path == 5) ||
                           (prec == EGNN_PREC_F16C8 && c->last_path == 7);
    int hs_used = 1;
    if (half_path && f16_node == 2 && node_post_split_supported(qs)) {
      int rc = launch_node_post_bf16(qs, st, true, true, defer_finish, &hs_used);
      if (rc) return rc;
    } else if (prec == EGNN_PREC_F16 && c->last_path == 6 && f16_node == 1 && node_post_bf16_supported(q)) {
      q.w1h_bf16 = lp.w1h_f16; q.w2h_bf16p = lp.w2h_f16p;
      int rc = launch_node_post_bf16(q, st, true);
      if (rc) return rc;
    } else if (prec == EGNN_PREC_BF16 && node_post_bf16_supported(q)) {
      int rc = launch_node_post_bf16(q, st, false, false, defer_finish, &hs_used);
      if (rc) return rc;
    } else {
      if (q.sq_from_agg) {   // the fp32 node kernel reads gscale
        hipLaunchKernelGGL(graph_sq_sums_kernel, dim3(per_graph ? c->B : 1), dim3(kThreads), 0, st, c->row_ptr, R, c->agg_x,
                           c->part_x, c->graph_ptr, N, per_graph, c->gscale);
        q.sq_from_agg = 0;
      }
      hipLaunchKernelGGL(node_post_kernel, dim3((N + kPostNodes - 1) / kPostNodes), dim3(kThreads),
                         post_smem_bytes(c->K1P, c->WhP), st, q);
    }
    if (defer_finish && hs_used > 1) {   // the next layer's node_pre (or launch_layer_begin's flush) adds the partials up
      c->pend.active = true; c->pend.hs = hs_used; c->pend.b2h = lp.b2h; c->pend.h_out = h_out;
    }
    prof_end(c, st);
    EGNN_HIP(hipGetLastError());
  }
  return EGNN_OK;
}

// defer_finish: only from a loop that launches the NEXT layer on h_out right away (egnn_forward, the sampler's step)
int launch_layer(egnn_ctx* c, hipStream_t st, int layer, int prec, int norm_scope, const float* h,
                 const float* x, float* h_out, float* x_out, bool need_gscale, bool defer_finish) {
  if (layer == 0) c->pend.active = false;   // (a call that failed half-way leaves nothing behind)
  static const bool defer_ok = !(getenv("EGNN_DEFER_FINISH") && atoi(getenv("EGNN_DEFER_FINISH")) == 0);   // A/B switch
  defer_finish = defer_finish && defer_ok;
  int rc = launch_layer_begin(c, st, layer, prec, norm_scope, h, x, need_gscale);
  if (rc) return rc;
  return launch_layer_end(c, st, layer, prec, norm_scope, h, x, h_out, x_out, defer_finish);
}

}  // namespace egnn

using namespace egnn;

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int egnn_create(egnn_ctx** out, int device) {
  if (!out) return EGNN_EINVAL;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
    set_error("no HIP device %d (count %d)", device, count);
    return EGNN_EHIP;
  }
  EGNN_HIP(hipSetDevice(device));
  {
    int rc = init_kernel_attributes();
    if (rc) return rc;
  }
  egnn_ctx* c = new egnn_ctx();
  c->device = device;
  *out = c;
  return EGNN_OK;
}

static void free_layer(LayerPack& lp) {
  void* ptrs[] = {lp.w1catT, lp.b1cat, lp.wdx, lp.wdm, lp.w2x_f32, lp.w2x_bf16, lp.b2x, lp.w3x, lp.w2m_f32,
                  lp.w2m_bf16, lp.b2m, lp.wa, lp.scal, lp.w1h_f32, lp.b1h, lp.w2h_f32, lp.b2h, lp.sc, lp.w2x_bf16s, lp.w2m_bf16s, lp.w1h_bf16, lp.w2h_bf16p,
                  lp.w2xT_bf16, lp.w2mT_bf16, lp.w1hl_bf16, lp.w2x_bf16s16, lp.w2x_bf16s_lo, lp.w2m_bf16s_lo,
                  lp.w2x_f16s16, lp.w2m_f16s, lp.w1h_f16, lp.w2h_f16p, lp.w1h_f16k, lp.w1h_f16k_lo, lp.w2h_f16p_lo,
                  lp.w2m_bf16s16, lp.w2m_f16s16, lp.w2x_c8, lp.w2m_c8, lp.c8_exp, lp.w2x_f16s, lp.w2x_c8w, lp.w2m_c8w};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  lp = LayerPack();
}

void sampler_free(egnn_ctx* c);

int egnn_destroy(egnn_ctx* c) {
  if (!c) return EGNN_EINVAL;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto& lp : c->layers) free_layer(lp);
  void* ptrs[] = {c->table, c->agg_m, c->agg_x, c->part_m, c->part_x, c->node_d2, c->gscale, c->bwd_s, c->h_partial,
                  c->h_tmp[0], c->h_tmp[1], c->x_tmp[0], c->x_tmp[1]};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  sampler_free(c);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  c->side = nullptr;   // the caller's stream (egnn_set_side_stream)
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  delete c;
  return EGNN_OK;
}

int egnn_set_model(egnn_ctx* c, int L, int H, int M, int Wm, int Wx, int Wh) {
  if (!c) return EGNN_EINVAL;
  ModelDims md;
  {
    const int rc = model_dims(L, H, M, Wm, Wx, Wh, &md);   // validation + padded widths (host_logic.cpp)
    if (rc) return rc;
  }
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto& lp : c->layers) free_layer(lp);
  c->layers.assign(L, LayerPack());
  c->L = L; c->H = H; c->M = M; c->Wm = Wm; c->Wx = Wx; c->Wh = Wh;
  c->WxP = md.WxP; c->WmP = md.WmP; c->MP = md.MP; c->cbx = md.cbx; c->cbm = md.cbm;
  c->WhP = md.WhP; c->HP = md.HP; c->K1P = md.K1P; c->K1Q = md.K1Q; c->TC = md.TC;
  c->cap_nodes = c->cap_tiles = c->cap_graphs = 0;  // MP / TC may have changed
  if (post_smem_bytes(c->K1P, c->WhP) > 160 * 1024 || edge_smem_bytes(64, c->MP) > 160 * 1024) {
    set_error("model does not fit the 160 KiB LDS budget");
    return EGNN_EINVAL;
  }
  return EGNN_OK;
}

int egnn_pack_layer(egnn_ctx* c, void* stream, int l, const float* m0_w, const float* m0_b, const float* m2_w,
                    const float* m2_b, const float* x0_w, const float* x0_b, const float* x2_w,
                    const float* x2_b, const float* x4_w, const float* x4_b, const float* h0_w,
                    const float* h0_b, const float* h2_w, const float* h2_b, const float* a_w,
                    const float* a_b) {
  if (!c || c->L == 0) { set_error("egnn_set_model first"); return EGNN_ESTATE; }
  if (l < 0 || l >= c->L) { set_error("layer %d out of range", l); return EGNN_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  EGNN_HIP(hipSetDevice(c->device));
  LayerPack& lp = c->layers[l];
  const int H = c->H, M = c->M, Wm = c->Wm, Wx = c->Wx, Wh = c->Wh;
  const int WxP = c->WxP, WmP = c->WmP, MP = c->MP, WhP = c->WhP, HP = c->HP, K1P = c->K1P, TC = c->TC;
  int rc;
  if (!lp.w1catT) {
    if ((rc = dev_alloc(&lp.w1catT, (size_t)H * TC))) return rc;
    if ((rc = dev_alloc(&lp.b1cat, (size_t)TC))) return rc;
    if ((rc = dev_alloc(&lp.wdx, (size_t)WxP))) return rc;
    if ((rc = dev_alloc(&lp.wdm, (size_t)WmP))) return rc;
    if ((rc = dev_alloc(&lp.w2x_f32, (size_t)WxP * WxP))) return rc;
    __bf16* tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)WxP * WxP))) return rc;
    lp.w2x_bf16 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&lp.b2x, (size_t)WxP))) return rc;
    if ((rc = dev_alloc(&lp.w3x, (size_t)WxP))) return rc;
    if ((rc = dev_alloc(&lp.w2m_f32, (size_t)MP * WmP))) return rc;
    if ((rc = dev_alloc(&tmp, (size_t)MP * WmP))) return rc;
    lp.w2m_bf16 = tmp;
    if ((rc = dev_alloc(&lp.b2m, (size_t)MP))) return rc;
    if ((rc = dev_alloc(&lp.wa, (size_t)MP))) return rc;
    if ((rc = dev_alloc(&lp.scal, (size_t)4))) return rc;
    if ((rc = dev_alloc(&lp.w1h_f32, (size_t)WhP * K1P))) return rc;
    if ((rc = dev_alloc(&lp.b1h, (size_t)WhP))) return rc;
    if ((rc = dev_alloc(&lp.w2h_f32, (size_t)HP * WhP))) return rc;
    if ((rc = dev_alloc(&lp.b2h, (size_t)HP))) return rc;
    if ((rc = dev_alloc(&lp.sc, (size_t)(H + 1) * TC + 3 * (size_t)WxP + WmP + 2 * (size_t)MP))) return rc;
    tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)WhP * c->K1Q))) return rc;
    lp.w1h_bf16 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)HP * WhP))) return rc;
    lp.w2h_bf16p = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)WxP * WxP))) return rc;
    lp.w2x_bf16s = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)MP * WmP))) return rc;
    lp.w2m_bf16s = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)WxP * WxP))) return rc;
    lp.w2x_bf16s16 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)WxP * WxP))) return rc;
    lp.w2x_bf16s_lo = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)MP * WmP))) return rc;
    lp.w2m_bf16s_lo = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)WxP * WxP))) return rc;
    lp.w2xT_bf16 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)MP * WmP))) return rc;
    lp.w2mT_bf16 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)(TC / 32) * 3 * 2 * 512))) return rc;
    lp.w1hl_bf16 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)MP * WmP))) return rc;
    lp.w2m_bf16s16 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)MP * WmP))) return rc;
    lp.w2m_f16s16 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)WxP * WxP))) return rc;      // fp16 streams: same sizes as their bf16 twins
    lp.w2x_f16s16 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)MP * WmP))) return rc;
    lp.w2m_f16s = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)WhP * c->K1Q))) return rc;
    lp.w1h_f16 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)HP * WhP))) return rc;
    lp.w2h_f16p = tmp; tmp = nullptr;
    if (HP <= 64 && H + MP > node_post_split_k() / 2 && H + MP <= node_post_split_k()) {   // shapes of the split-operand node MLP
      if ((rc = dev_alloc(&tmp, (size_t)WhP * node_post_split_k()))) return rc;
      lp.w1h_f16k = tmp; tmp = nullptr;
      if ((rc = dev_alloc(&tmp, (size_t)WhP * node_post_split_k()))) return rc;
      lp.w1h_f16k_lo = tmp; tmp = nullptr;
      if ((rc = dev_alloc(&tmp, (size_t)HP * WhP))) return rc;
      lp.w2h_f16p_lo = tmp;
    }
    // precision f16c8: e4m3 fragments of heads + remainders (2 bytes per weight) and the block-scale exponents
    tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)WxP * WxP))) return rc;
    lp.w2x_c8 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)MP * WmP))) return rc;
    lp.w2m_c8 = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)WxP * WxP))) return rc;
    lp.w2x_f16s = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)WxP * WxP))) return rc;
    lp.w2x_c8w = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&tmp, (size_t)MP * WmP))) return rc;
    lp.w2m_c8w = tmp; tmp = nullptr;
    if ((rc = dev_alloc(&lp.c8_exp, (size_t)8))) return rc;   // [0..3] scale exponents {x: hi, lo, m: hi, lo}, [4..5] max |w| scratch
  }
  const dim3 g(256), b(256);
  hipLaunchKernelGGL(pack_first, g, b, 0, st, x0_w, x0_b, m0_w, m0_b, H, Wx, Wm, WxP, WmP, lp.w1catT, lp.b1cat);
  hipLaunchKernelGGL(pad_copy, dim3(8), b, 0, st, x0_w + 2 * H, Wx, 2 * H + 1, lp.wdx, WxP);
  hipLaunchKernelGGL(pad_copy, dim3(8), b, 0, st, m0_w + 2 * H, Wm, 2 * H + 1, lp.wdm, WmP);
  hipLaunchKernelGGL(pack_frags_f32, g, b, 0, st, x2_w, Wx, Wx, Wx, WxP, WxP, lp.w2x_f32);
  hipLaunchKernelGGL(pack_frags_bf16<__bf16>, g, b, 0, st, x2_w, Wx, Wx, Wx, WxP, WxP, reinterpret_cast<__bf16*>(lp.w2x_bf16), 1.0f);
  hipLaunchKernelGGL(pad_copy, dim3(8), b, 0, st, x2_b, Wx, 1, lp.b2x, WxP);
  hipLaunchKernelGGL(pad_copy, dim3(8), b, 0, st, x4_w, Wx, 1, lp.w3x, WxP);
  hipLaunchKernelGGL(pack_frags_f32, g, b, 0, st, m2_w, M, Wm, Wm, MP, WmP, lp.w2m_f32);
  hipLaunchKernelGGL(pack_frags_bf16<__bf16>, g, b, 0, st, m2_w, M, Wm, Wm, MP, WmP, reinterpret_cast<__bf16*>(lp.w2m_bf16), 1.0f);
  hipLaunchKernelGGL(pad_copy, dim3(8), b, 0, st, m2_b, M, 1, lp.b2m, MP);
  hipLaunchKernelGGL(pad_copy, dim3(8), b, 0, st, a_w, M, 1, lp.wa, MP);
  hipLaunchKernelGGL(pad_copy, dim3(1), dim3(64), 0, st, x4_b, 1, 1, lp.scal, 1);
  hipLaunchKernelGGL(pad_copy, dim3(1), dim3(64), 0, st, a_b, 1, 1, lp.scal + 1, 1);
  // mlp_h.0 sees [h | sum_m]; the kernel's K index is [h (H) | sum_m (MP, zero-padded beyond M)]
  hipLaunchKernelGGL(pack_frags_f32, g, b, 0, st, h0_w, Wh, H + M, H + M, WhP, K1P, lp.w1h_f32);
  hipLaunchKernelGGL(pad_copy, dim3(8), b, 0, st, h0_b, Wh, 1, lp.b1h, WhP);
  hipLaunchKernelGGL(pack_frags_f32, g, b, 0, st, h2_w, H, Wh, Wh, HP, WhP, lp.w2h_f32);
  hipLaunchKernelGGL(pad_copy, dim3(8), b, 0, st, h2_b, H, 1, lp.b2h, HP);
  {  // scaled copies for the bf16 fast path (see LayerPack::sc)
    float* o = lp.sc;
    const float s1 = kNegLog2e, s2 = kNegInvLog2e;
    hipLaunchKernelGGL(scale_copy, g, b, 0, st, lp.w1catT, (size_t)H * TC, s1, o); o += (size_t)H * TC;
    hipLaunchKernelGGL(scale_copy, dim3(8), b, 0, st, lp.b1cat, (size_t)TC, s1, o); o += TC;
    hipLaunchKernelGGL(scale_copy, dim3(8), b, 0, st, lp.wdx, (size_t)WxP, s1, o); o += WxP;
    hipLaunchKernelGGL(scale_copy, dim3(8), b, 0, st, lp.wdm, (size_t)WmP, s1, o); o += WmP;
    hipLaunchKernelGGL(scale_copy, dim3(8), b, 0, st, lp.b2x, (size_t)WxP, s1, o); o += WxP;
    hipLaunchKernelGGL(scale_copy, dim3(8), b, 0, st, lp.w3x, (size_t)WxP, s2, o); o += WxP;
    hipLaunchKernelGGL(scale_copy, dim3(8), b, 0, st, lp.b2m, (size_t)MP, s1, o); o += MP;
    hipLaunchKernelGGL(scale_copy, dim3(8), b, 0, st, lp.wa, (size_t)MP, s2, o);
    hipLaunchKernelGGL(pack_frags_bf16<__bf16>, g, b, 0, st, x2_w, Wx, Wx, Wx, WxP, WxP, reinterpret_cast<__bf16*>(lp.w2x_bf16s), s2);
    hipLaunchKernelGGL(pack_frags_bf16<__bf16>, g, b, 0, st, m2_w, M, Wm, Wm, MP, WmP, reinterpret_cast<__bf16*>(lp.w2m_bf16s), s2);
    hipLaunchKernelGGL(pack_frags_bf16_n16<__bf16>, g, b, 0, st, x2_w, Wx, Wx, Wx, WxP, WxP, reinterpret_cast<__bf16*>(lp.w2x_bf16s16), s2);
    hipLaunchKernelGGL(pack_frags_bf16_n16<__bf16>, g, b, 0, st, m2_w, M, Wm, Wm, MP, WmP, reinterpret_cast<__bf16*>(lp.w2m_bf16s16), s2);
    hipLaunchKernelGGL(pack_frags_bf16_n16<_Float16>, g, b, 0, st, m2_w, M, Wm, Wm, MP, WmP, reinterpret_cast<_Float16*>(lp.w2m_f16s16), s2 * kF16WScale);
    // precision fp16: the same streams as fp16 fragments, times 2^8 (kernels.h "MFMA operand type")
    hipLaunchKernelGGL(pack_frags_bf16_n16<_Float16>, g, b, 0, st, x2_w, Wx, Wx, Wx, WxP, WxP, reinterpret_cast<_Float16*>(lp.w2x_f16s16), s2 * kF16WScale);
    hipLaunchKernelGGL(pack_frags_bf16<_Float16>, g, b, 0, st, m2_w, M, Wm, Wm, MP, WmP, reinterpret_cast<_Float16*>(lp.w2m_f16s), s2 * kF16WScale);
    hipLaunchKernelGGL(pack_frags_bf16<_Float16>, g, b, 0, st, h0_w, Wh, H + M, H + M, WhP, c->K1Q, reinterpret_cast<_Float16*>(lp.w1h_f16), kF16WScale);
    hipLaunchKernelGGL(pack_frags_bf16_accperm<_Float16>, g, b, 0, st, h2_w, H, Wh, Wh, HP, WhP, reinterpret_cast<_Float16*>(lp.w2h_f16p), kF16WScale);
    if (lp.w1h_f16k) {   // split-operand node MLP: heads + remainders, mlp_h.0 with K padded to the ring's two turns
      const int KS = node_post_split_k();
      hipLaunchKernelGGL(pack_frags_bf16<_Float16>, g, b, 0, st, h0_w, Wh, H + M, H + M, WhP, KS, reinterpret_cast<_Float16*>(lp.w1h_f16k), kF16WScale);
      hipLaunchKernelGGL((pack_frags_bf16<_Float16, true>), g, b, 0, st, h0_w, Wh, H + M, H + M, WhP, KS, reinterpret_cast<_Float16*>(lp.w1h_f16k_lo), kF16WScale);
      hipLaunchKernelGGL((pack_frags_bf16_accperm<_Float16, true>), g, b, 0, st, h2_w, H, Wh, Wh, HP, WhP, reinterpret_cast<_Float16*>(lp.w2h_f16p_lo), kF16WScale);
    }
    // precision f16c8: the same scaled weights as e4m3 head / remainder fragments for the block-scaled correction product
    if ((rc = pack_c8_stream(x2_w, Wx, Wx, Wx, WxP, WxP, lp.w2x_c8, s2 * kF16WScale, lp.c8_exp, reinterpret_cast<unsigned*>(lp.c8_exp + 4), st))) return rc;
    if ((rc = pack_c8_stream(m2_w, M, Wm, Wm, MP, WmP, lp.w2m_c8, s2 * kF16WScale, lp.c8_exp + 2, reinterpret_cast<unsigned*>(lp.c8_exp + 5), st))) return rc;
    // ... and for the 32x32 tiles of edge_f16c8w.hip (same scale exponents)
    hipLaunchKernelGGL(pack_frags_bf16<_Float16>, g, b, 0, st, x2_w, Wx, Wx, Wx, WxP, WxP, reinterpret_cast<_Float16*>(lp.w2x_f16s), s2 * kF16WScale);
    if ((rc = pack_c8w_stream(x2_w, Wx, Wx, Wx, WxP, WxP, lp.w2x_c8w, s2 * kF16WScale, lp.c8_exp, st))) return rc;
    if ((rc = pack_c8w_stream(m2_w, M, Wm, Wm, MP, WmP, lp.w2m_c8w, s2 * kF16WScale, lp.c8_exp + 2, st))) return rc;
    hipLaunchKernelGGL(pack_frags_bf16_lo, g, b, 0, st, x2_w, Wx, Wx, Wx, WxP, WxP, reinterpret_cast<__bf16*>(lp.w2x_bf16s_lo), s2);
    hipLaunchKernelGGL(pack_frags_bf16_lo, g, b, 0, st, m2_w, M, Wm, Wm, MP, WmP, reinterpret_cast<__bf16*>(lp.w2m_bf16s_lo), s2);
    hipLaunchKernelGGL(pack_frags_bf16<__bf16>, g, b, 0, st, h0_w, Wh, H + M, H + M, WhP, c->K1Q, reinterpret_cast<__bf16*>(lp.w1h_bf16), 1.0f);
    hipLaunchKernelGGL(pack_frags_bf16_accperm<__bf16>, g, b, 0, st, h2_w, H, Wh, Wh, HP, WhP, reinterpret_cast<__bf16*>(lp.w2h_bf16p), 1.0f);
    if (H <= 48)   // hi/lo bf16 fragments of the scaled first-layer weights (node_pre_hilo_kernel)
      hipLaunchKernelGGL(pack_w1_hilo, g, b, 0, st, lp.sc, H, TC, reinterpret_cast<__bf16*>(lp.w1hl_bf16));
    // transposed packs for the backward dgrad: B[k = second-layer output][column = hidden unit]
    hipLaunchKernelGGL(pack_frags_bf16_T, g, b, 0, st, x2_w, Wx, Wx, Wx, WxP, WxP, reinterpret_cast<__bf16*>(lp.w2xT_bf16));
    hipLaunchKernelGGL(pack_frags_bf16_T, g, b, 0, st, m2_w, M, Wm, Wm, WmP, MP, reinterpret_cast<__bf16*>(lp.w2mT_bf16));
  }
  EGNN_HIP(hipGetLastError());
  lp.packed = true;
  return EGNN_OK;
}

int egnn_set_side_stream(egnn_ctx* c, void* stream) {
  if (!c) return EGNN_EINVAL;
  c->side = reinterpret_cast<hipStream_t>(stream);
  if (c->smp.ready) {   // captured graphs carry the fork / join edges of the old setting
    for (int i = 0; i < 2; ++i) {
      if (c->smp.graph_exec[i]) { (void)hipGraphExecDestroy(c->smp.graph_exec[i]); c->smp.graph_exec[i] = nullptr; }
      if (c->smp.graph[i]) { (void)hipGraphDestroy(c->smp.graph[i]); c->smp.graph[i] = nullptr; }
    }
    c->smp.graph_prec = c->smp.graph_norm = -1;
  }
  return fork_streams(c);
}

int egnn_set_graph(egnn_ctx* c, int N, int E, int B, const int32_t* edge_dst, const int32_t* edge_src,
                   const int32_t* row_ptr, const int32_t* graph_ptr, const int32_t* node_graph) {
  if (!c) return EGNN_EINVAL;
  {
    const int rc = graph_args_check(c->L != 0, N, E, B, edge_dst, edge_src, row_ptr, graph_ptr, node_graph);
    if (rc) return rc;
  }
  EGNN_HIP(hipSetDevice(c->device));
  c->N = N; c->E = E; c->B = B;
  c->edge_dst = edge_dst; c->edge_src = edge_src; c->row_ptr = row_ptr; c->graph_ptr = graph_ptr;
  c->node_graph = node_graph;
  c->smp.ready = false;
  return reserve(c);
}

static int check_ready(egnn_ctx* c, int prec, int norm_scope) {
  return precision_scope_check(c && c->L != 0 && c->N != 0, prec, norm_scope);
}

int egcl_forward(egnn_ctx* c, void* stream, int layer, int prec, int norm_scope, const float* h, const float* x,
                 float* h_out, float* x_out) {
  int rc = check_ready(c, prec, norm_scope);
  if (rc) return rc;
  if (layer < 0 || layer >= c->L || !h || !x || !h_out || !x_out || h == h_out || x == x_out) {
    set_error("bad egcl_forward arguments");
    return EGNN_EINVAL;
  }
  // the single-layer entry keeps the per-graph sums of d^2 available for egcl_read_aggregates
  return launch_layer(c, reinterpret_cast<hipStream_t>(stream), layer, prec, norm_scope, h, x, h_out, x_out, true);
}

int egcl_forward_save(egnn_ctx* c, void* stream, int layer, int norm_scope, const float* h, const float* x, float* h_out,
                      float* x_out, void* s1x, void* s1m, void* t2x, void* t2m, float* s_shares) {
  int rc = check_ready(c, EGNN_PREC_BF16, norm_scope);
  if (rc) return rc;
  if (layer < 0 || layer >= c->L || !h || !x || !h_out || !x_out || h == h_out || x == x_out || !s1x || !s1m || !t2x || !t2m ||
      !s_shares) {
    set_error("bad egcl_forward_save arguments");
    return EGNN_EINVAL;
  }
  if (c->E == 0 || !backward_recompute_supported(c)) { set_error("egcl_forward_save is not available for these widths"); return EGNN_EINVAL; }
  if ((size_t)c->E * c->WxP * 2 >= ((size_t)1 << 32)) {   // the activation chunks are stored through a 32-bit buffer descriptor
    set_error("egcl_forward_save: E x Wx x 2 bytes must stay below 4 GiB (%d edges); train on smaller batches or use the recompute path", c->E);
    return EGNN_EINVAL;
  }
  c->save_s1x = s1x; c->save_s1m = s1m; c->save_t2x = t2x; c->save_t2m = t2m; c->save_s = s_shares;
  rc = launch_layer(c, reinterpret_cast<hipStream_t>(stream), layer, EGNN_PREC_BF16, norm_scope, h, x, h_out, x_out, true);
  c->save_s1x = c->save_s1m = c->save_t2x = c->save_t2m = nullptr; c->save_s = nullptr;
  return rc;
}

int egcl_forward_begin(egnn_ctx* c, void* stream, int layer, int prec, int norm_scope, const float* h, const float* x,
                       float* d_sq_sums) {
  int rc = check_ready(c, prec, norm_scope);
  if (rc) return rc;
  if (layer < 0 || layer >= c->L || !h || !x) { set_error("bad egcl_forward_begin arguments"); return EGNN_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if ((rc = launch_layer_begin(c, st, layer, prec, norm_scope, h, x, true))) return rc;
  if (d_sq_sums)
    EGNN_HIP(hipMemcpyAsync(d_sq_sums, c->gscale, sizeof(float) * (norm_scope == EGNN_NORM_GRAPH ? c->B : 1),
                            hipMemcpyDeviceToDevice, st));
  return EGNN_OK;
}

int egcl_forward_end(egnn_ctx* c, void* stream, int layer, int prec, int norm_scope, const float* h, const float* x,
                     const float* d_sq_sums, float* h_out, float* x_out) {
  int rc = check_ready(c, prec, norm_scope);
  if (rc) return rc;
  if (layer < 0 || layer >= c->L || !h || !x || !h_out || !x_out || h == h_out || x == x_out) {
    set_error("bad egcl_forward_end arguments");
    return EGNN_EINVAL;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d_sq_sums)
    EGNN_HIP(hipMemcpyAsync(c->gscale, d_sq_sums, sizeof(float) * (norm_scope == EGNN_NORM_GRAPH ? c->B : 1),
                            hipMemcpyDeviceToDevice, st));
  c->sq_from_agg = false;
  return launch_layer_end(c, st, layer, prec, norm_scope, h, x, h_out, x_out);
}

int egnn_forward(egnn_ctx* c, void* stream, int prec, int norm_scope, const float* h, const float* x, float* h_out,
                 float* x_out) {
  int rc = check_ready(c, prec, norm_scope);
  if (rc) return rc;
  if (!h || !x || !h_out || !x_out || h == h_out || x == x_out) { set_error("bad egnn_forward arguments"); return EGNN_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float *hc = h, *xc = x;
  for (int l = 0; l < c->L; ++l) {
    float* ho = (l == c->L - 1) ? h_out : c->h_tmp[l & 1];
    float* xo = (l == c->L - 1) ? x_out : c->x_tmp[l & 1];
    if ((rc = launch_layer(c, st, l, prec, norm_scope, hc, xc, ho, xo, false, l + 1 < c->L))) return rc;
    hc = ho; xc = xo;
  }
  return EGNN_OK;
}

int egnn_debug_stamps(egnn_ctx* c, unsigned long long* host_out) {
  if (!c || !c->stamps || !host_out) return EGNN_EINVAL;
  EGNN_HIP(hipDeviceSynchronize());
  EGNN_HIP(hipMemcpy(host_out, c->stamps, 2 * 8 * 32 * 4 * 8, hipMemcpyDeviceToHost));
  return EGNN_OK;
}

int egnn_profile_enable(egnn_ctx* c, int enable) {
  if (!c) return EGNN_EINVAL;
  c->prof = enable != 0;
  c->ev_used = 0;
  return EGNN_OK;
}

int egnn_profile_read(egnn_ctx* c, float* edge_ms_avg, int* edge_launches, float* node_ms_avg) {
  if (!c) return EGNN_EINVAL;
  EGNN_HIP(hipDeviceSynchronize());
  double se = 0, sn = 0;
  int ne = 0, nn = 0;
  for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) != hipSuccess) continue;
    if (c->ev_kind[i / 2] == 0) { se += ms; ++ne; } else { sn += ms; ++nn; }
  }
  if (edge_ms_avg) *edge_ms_avg = ne ? (float)(se / ne) : 0.f;
  if (edge_launches) *edge_launches = ne;
  if (node_ms_avg) *node_ms_avg = nn ? (float)(sn / nn) : 0.f;
  c->ev_used = 0;
  return EGNN_OK;
}

}  // extern "C"
