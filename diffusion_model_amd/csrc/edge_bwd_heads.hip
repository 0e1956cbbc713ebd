// Training backward from SAVED second-layer pre-activations (gfx950): dL/d(a2) of both edge MLPs without a recompute
// pass over the edges.
//
// egcl_forward_save (edge kernels instantiated with SAVE) leaves, per layer, t2 = -log2(e) * (a2 + b2) as bf16 for the
// coordinate MLP ([E][Wx]) and the message MLP ([E][M]) in HBM.  The two kernels here turn a chunk of those rows into
// dL/d(a2) IN PLACE -- the same arithmetic as the BWD epilogues of edge_bf16_v3.hip / edge_bf16_v4.hip, which produce it
// from the accumulators of a recomputed forward -- plus the bias / w3 / wa column sums and dL/d(b3), dL/d(ba):
//
//   coordinate head (EquivariantGraphNeuralNetwork.py:62-65): s = w3 . SiLU(a2) + b3, xm = (x_i - x_j) * s
//       dL/ds_e = dL/d(sum_x[i]) . (x_i - x_j),  dL/da2[e][n] = dL/ds_e * w3[n] * SiLU'(a2[e][n])
//   message head (:57-60): m = SiLU(a2), z = wa . m + ba, gate = sigmoid(z), out = m * gate; with g = dL/d(sum_m[i]):
//       dL/dm = g * gate + (g . m) gate (1 - gate) wa,  dL/da2 = dL/dm * SiLU'(a2)
//
// Both are HBM-bound element-wise passes (read + write one bf16 per element, 12-15 vector instructions each): 2.1 GB per
// 2^19-edge chunk of the coordinate MLP against the 1.6 ms the recompute kernel needs for the same chunk.
#include "kernels.h"

namespace egnn {

namespace {

constexpr int kHT = 256;       // threads per workgroup
constexpr int kHRows = 512;    // edge rows per workgroup (column sums leave as one atomic per column and workgroup)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// ---- coordinate head: W columns per row, 8 consecutive columns per thread --------------------------------------------
template <int W>
__global__ __launch_bounds__(kHT) void heads_x_kernel(int n_edges, const int* __restrict__ dst, const int* __restrict__ src,
                                                      const float* __restrict__ x, const float* __restrict__ g_sum_x,
                                                      const float* __restrict__ w3s,   // w3 * (-1 / log2 e) (packed vector)
                                                      __bf16* __restrict__ t2,         // [n_edges][W] in: t2, out: dL/da2
                                                      float* __restrict__ g_b2, float* __restrict__ g_w3,
                                                      float* __restrict__ g_b3) {
  constexpr int TPR = W / 8;          // threads per row
  constexpr int RPP = kHT / TPR;      // rows per pass
  __shared__ float s_gsc[kHRows];
  __shared__ float s_red[kHT / 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int e_base = blockIdx.x * kHRows;
  // dL/ds_e of this workgroup's rows, one row per thread and sweep; their sum is this workgroup's share of dL/db3
  float gb3 = 0.f;
  for (int rr = tid; rr < kHRows; rr += kHT) {
    const int e = e_base + rr;
    float v = 0.f;
    if (e < n_edges) {
      const int i = dst[e], j = src[e];
      v = g_sum_x[3 * i] * (x[3 * i] - x[3 * j]) + g_sum_x[3 * i + 1] * (x[3 * i + 1] - x[3 * j + 1]) +
          g_sum_x[3 * i + 2] * (x[3 * i + 2] - x[3 * j + 2]);
    }
    s_gsc[rr] = v;
    gb3 += v;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) gb3 += __shfl_xor(gb3, m);
  if (lane == 0) s_red[tid >> 6] = gb3;
  __syncthreads();
  if (tid == 0) {
    float v = 0.f;
    for (int w = 0; w < kHT / 64; ++w) v += s_red[w];
    atomicAdd(g_b3, v);
  }
  const int c0 = 8 * (tid % TPR), r0 = tid / TPR;
  float w3n[8], cs_b[8], cs_w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { w3n[k] = w3s[c0 + k] * kNegLog2e; cs_b[k] = 0.f; cs_w[k] = 0.f; }
  const int rows = min(kHRows, n_edges - e_base);
  for (int rr = r0; rr < rows; rr += RPP) {
    __bf16* ptr = t2 + (size_t)(e_base + rr) * W + c0;
    const bf16x8_t tv = *reinterpret_cast<const bf16x8_t*>(ptr);
    const float gsc = s_gsc[rr];
    bf16x8_t gv;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float sv, ds;
      silu_grad_s((float)tv[k], sv, ds);
      const float g = gsc * w3n[k] * ds;
      cs_b[k] += g;
      cs_w[k] = fmaf(gsc, sv, cs_w[k]);
      gv[k] = (__bf16)g;
    }
    *reinterpret_cast<bf16x8_t*>(ptr) = gv;
  }
  // column sums: the RPP row groups of this workgroup first (through LDS), then one atomic per column
  __shared__ float s_cs[2][RPP][W];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s_cs[0][r0][c0 + k] = cs_b[k]; s_cs[1][r0][c0 + k] = cs_w[k]; }
  __syncthreads();
  for (int c = tid; c < W; c += kHT) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int g = 0; g < RPP; ++g) { a += s_cs[0][g][c]; b += s_cs[1][g][c]; }
    atomicAdd(g_b2 + c, a);
    atomicAdd(g_w3 + c, b);
  }
}

// ---- message head: one wave per row, 4 consecutive columns per lane (M = 256) ------------------------------------------
__global__ __launch_bounds__(kHT) void heads_m_kernel(int n_edges, const int* __restrict__ dst,
                                                      const float* __restrict__ g_sum_m,   // [N][256]
                                                      const float* __restrict__ was,        // wa * (-1 / log2 e)
                                                      const float* __restrict__ scal,       // scal[1] = ba
                                                      __bf16* __restrict__ t2,              // [n_edges][256] in / out
                                                      float* __restrict__ g_b2, float* __restrict__ g_wa,
                                                      float* __restrict__ g_ba) {
  constexpr int M = 256;
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e_base = blockIdx.x * kHRows;
  const int rows = min(kHRows, n_edges - e_base);
  const int c0 = 4 * lane;
  float wan[4], cs_b[4], cs_w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { wan[k] = was[c0 + k] * kNegLog2e; cs_b[k] = 0.f; cs_w[k] = 0.f; }
  const float ba = scal[1];
  float gba = 0.f;
  // two rows per sweep: their two pairs of wave reductions (12 dependent cross-lane steps each) overlap
  for (int rr = 2 * wave; rr < rows; rr += 2 * (kHT / 64)) {
    const bool two = rr + 1 < rows;
    bf16x4_t* row[2];
    bf16x4_t tv[2];
    f32x4 g[2];
    float m[2][4], ds[2][4], z[2] = {0.f, 0.f}, d[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = e_base + rr + ((u == 1 && two) ? 1 : 0);
      row[u] = reinterpret_cast<bf16x4_t*>(t2 + (size_t)e * M + c0);
      tv[u] = *row[u];
      g[u] = *reinterpret_cast<const f32x4*>(g_sum_m + (size_t)dst[e] * M + c0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        silu_grad_s((float)tv[u][k], m[u][k], ds[u][k]);
        z[u] = fmaf(wan[k], m[u][k], z[u]);
        d[u] = fmaf(g[u][k], m[u][k], d[u]);
      }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      z[0] += __shfl_xor(z[0], s); d[0] += __shfl_xor(d[0], s);
      z[1] += __shfl_xor(z[1], s); d[1] += __shfl_xor(d[1], s);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      const float gate = sigmoid_f(z[u] + ba);
      const float coef = d[u] * gate * (1.0f - gate);
      gba += coef;
      bf16x4_t gv;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float ga = fmaf(g[u][k], gate, coef * wan[k]) * ds[u][k];
        cs_b[k] += ga;
        cs_w[k] = fmaf(coef, m[u][k], cs_w[k]);
        gv[k] = (__bf16)ga;
      }
      *row[u] = gv;
    }
  }
  __shared__ float s_cs[2][kHT / 64][M];
  __shared__ float s_red[kHT / 64];
#pragma unroll
  for (int k = 0; k < 4; ++k) { s_cs[0][wave][c0 + k] = cs_b[k]; s_cs[1][wave][c0 + k] = cs_w[k]; }
  if (lane == 0) s_red[wave] = gba;   // every lane of a wave holds the same coef
  __syncthreads();
  for (int c = tid; c < M; c += kHT) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < kHT / 64; ++w) { a += s_cs[0][w][c]; b += s_cs[1][w][c]; }
    atomicAdd(g_b2 + c, a);
    atomicAdd(g_wa + c, b);
  }
  if (tid == 0) {
    float v = 0.f;
    for (int w = 0; w < kHT / 64; ++w) v += s_red[w];
    atomicAdd(g_ba, v);
  }
}

}  // namespace

bool heads_saved_supported(int WxP, int MP) { return (WxP == 256 || WxP == 512 || WxP == 1024) && MP == 256; }

int launch_heads_saved(int n_edges, const int* dst, const int* src, const float* x, const float* g_sum_x, const float* g_sum_m,
                       int WxP, int MP, const float* w3s, const float* was, const float* scal, void* t2x, void* t2m,
                       float* g_b2x, float* g_w3, float* g_b3, float* g_b2m, float* g_wa, float* g_ba, hipStream_t st) {
  if (n_edges <= 0) return EGNN_OK;
  const dim3 grid((n_edges + kHRows - 1) / kHRows), block(kHT);
  __bf16* tx = static_cast<__bf16*>(t2x);
  if (WxP == 1024) hipLaunchKernelGGL(heads_x_kernel<1024>, grid, block, 0, st, n_edges, dst, src, x, g_sum_x, w3s, tx, g_b2x, g_w3, g_b3);
  else if (WxP == 512) hipLaunchKernelGGL(heads_x_kernel<512>, grid, block, 0, st, n_edges, dst, src, x, g_sum_x, w3s, tx, g_b2x, g_w3, g_b3);
  else hipLaunchKernelGGL(heads_x_kernel<256>, grid, block, 0, st, n_edges, dst, src, x, g_sum_x, w3s, tx, g_b2x, g_w3, g_b3);
  EGNN_HIP(hipGetLastError());
  hipLaunchKernelGGL(heads_m_kernel, grid, block, 0, st, n_edges, dst, g_sum_m, was, scal, static_cast<__bf16*>(t2m), g_b2m, g_wa,
                     g_ba);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace egnn
