// Backward stage kernels of one EGCL layer (training, reference EquivariantGraphNeuralNetwork.py:55-71 under
// autograd).  The backward of a layer is a chain
//     first-layer activations -> [GEMM] -> heads -> [GEMM x2] -> first-layer gradient -> [GEMM x2]
// in which the bracketed steps are plain dense GEMMs (dgrad / wgrad of mlp_x.2, mlp_m.2 and of the two first
// Linear layers) that the host runs with the BLAS library, and everything between them -- gathers, SiLU and its
// derivative, the scalar heads (mlp_x.4, the attention gate), row reductions and the column-sum gradients of
// biases / w3 / wa -- is fused into the three kernels below, one HBM pass each, in place on the GEMM buffers.
// Storage type of the [edges, W] buffers follows the precision (float or bf16); all arithmetic is fp32.
#include "kernels.h"

namespace egnn {
namespace {

template <typename T> struct Io;
template <> struct Io<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Io<__bf16> {
  static __device__ __forceinline__ float ld(const __bf16* p) { return (float)*p; }
  static __device__ __forceinline__ void st(__bf16* p, float v) { *p = (__bf16)v; }
};

// s = SiLU(a), ds = dSiLU/da = sig * (1 + a * (1 - sig))
__device__ __forceinline__ void silu_and_grad(float a, float& s, float& ds) {
  const float sg = sigmoid_f(a);
  s = a * sg;
  ds = sg * (1.0f + a * (1.0f - sg));
}

constexpr int kBwdRows = 32;   // edge rows per tile of the elementwise kernels

// ---- first-layer activations: out[e][c] = SiLU(P[dst e][c] + Q[src e][c] + wd[c] * d2[e]) ---------------------
// P = h . W1[:, :H]^T + b1 and Q = h . W1[:, H:2H]^T are per-NODE tables (:56's concatenation factorised);
// C = columns of the table (mlp_x.0 and mlp_m.0 side by side).  GRAD = false writes the activation, GRAD = true
// multiplies the buffer in place by SiLU'(pre-activation)  (dL/da1 = dL/ds1 * SiLU'(a1)).
template <typename T, bool GRAD>
__global__ __launch_bounds__(kThreads) void bwd_l1_kernel(int n_edges, int C, const int* __restrict__ dst,
                                                           const int* __restrict__ src, const float* __restrict__ P,
                                                           const float* __restrict__ Q, const float* __restrict__ wd,
                                                           const float* __restrict__ d2, T* __restrict__ buf) {
  const int e0 = blockIdx.x * kBwdRows, e1 = min(e0 + kBwdRows, n_edges);
  if ((C & 3) == 0) {
    for (int c = 4 * threadIdx.x; c < C; c += 4 * kThreads) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(wd + c);
      for (int e = e0; e < e1; ++e) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(P + (size_t)dst[e] * C + c);
        const f32x4 q = *reinterpret_cast<const f32x4*>(Q + (size_t)src[e] * C + c);
        const float d = d2[e];
        T* o = buf + (size_t)e * C + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s, ds;
          silu_and_grad(fmaf(w[j], d, p[j] + q[j]), s, ds);
          if (GRAD) Io<T>::st(o + j, Io<T>::ld(o + j) * ds);
          else Io<T>::st(o + j, s);
        }
      }
    }
  } else {
    for (int c = threadIdx.x; c < C; c += kThreads) {
      const float w = wd[c];
      for (int e = e0; e < e1; ++e) {
        float s, ds;
        silu_and_grad(fmaf(w, d2[e], P[(size_t)dst[e] * C + c] + Q[(size_t)src[e] * C + c]), s, ds);
        T* o = buf + (size_t)e * C + c;
        if (GRAD) Io<T>::st(o, Io<T>::ld(o) * ds);
        else Io<T>::st(o, s);
      }
    }
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- heads: everything between the second-layer GEMMs and their gradients ------------------------------------
// in : a2x[e][W] = s1x . W2x^T, a2m[e][M] = s1m . W2m^T (no bias), upstream gradients of the segment sums
// out: a2x <- dL/da2x, a2m <- dL/da2m (in place), g_diff[e][3] = dL/d(x_i - x_j) through the scalar s_e,
//      column sums into g_b2x, g_w3, g_b3, g_b2m, g_wa, g_ba (atomic adds, fp32)
// x branch (:62-65): s = w3 . SiLU(a2x + b2x) + b3,  xm = (x_i - x_j) * s,  dL/ds = g_agg_x[i] . (x_i - x_j)
// m branch (:57-60): m = SiLU(a2m + b2m), gate = sigmoid(wa . m + ba), out = m * gate, dL/dout = g_agg_m[i]
struct HeadsParams {
  int n_edges, W, M;
  const int *dst, *src;
  const float *x, *g_agg_x, *g_agg_m;
  void *a2x, *a2m;
  const float *b2x, *w3, *b3, *b2m, *wa, *ba;
  float *g_diff, *g_b2x, *g_w3, *g_b3, *g_b2m, *g_wa, *g_ba;
};

template <typename T>
__global__ __launch_bounds__(kThreads) void bwd_heads_kernel(const HeadsParams p) {
  extern __shared__ float cs[];   // column sums kept across the tiles of this workgroup: [b2x | w3 | b2m | wa]
  __shared__ float s_gsc[kBwdRows], s_gate[kBwdRows], s_coef[kBwdRows];
  __shared__ int s_dst[kBwdRows];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = p.W, M = p.M;
  T* a2x = static_cast<T*>(p.a2x);
  T* a2m = static_cast<T*>(p.a2m);
  for (int c = tid; c < 2 * W + 2 * M; c += kThreads) cs[c] = 0.f;
  float sum_gsc = 0.f, sum_coef = 0.f;
  const float b3 = p.b3[0], ba = p.ba[0];
  const int tiles = (p.n_edges + kBwdRows - 1) / kBwdRows;
  __syncthreads();
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int e0 = tile * kBwdRows, rows = min(kBwdRows, p.n_edges - e0);
    // phase 1: one wave per row, row reductions
    for (int r = wave; r < rows; r += kWaves) {
      const int e = e0 + r, i = p.dst[e], j = p.src[e];
      float acc = 0.f;
      for (int c = lane; c < W; c += 64) {
        const float a = Io<T>::ld(a2x + (size_t)e * W + c) + p.b2x[c];
        acc = fmaf(p.w3[c], silu_f(a), acc);
      }
      float z = 0.f, dot = 0.f;
      for (int c = lane; c < M; c += 64) {
        const float m = silu_f(Io<T>::ld(a2m + (size_t)e * M + c) + p.b2m[c]);
        z = fmaf(p.wa[c], m, z);
        dot = fmaf(p.g_agg_m[(size_t)i * M + c], m, dot);
      }
      acc = wave_sum(acc); z = wave_sum(z); dot = wave_sum(dot);
      if (lane == 0) {
        const float sc = acc + b3;
        const float dx = p.x[3 * i] - p.x[3 * j], dy = p.x[3 * i + 1] - p.x[3 * j + 1], dz = p.x[3 * i + 2] - p.x[3 * j + 2];
        const float gx = p.g_agg_x[3 * i], gy = p.g_agg_x[3 * i + 1], gz = p.g_agg_x[3 * i + 2];
        s_gsc[r] = gx * dx + gy * dy + gz * dz;
        p.g_diff[3 * (size_t)e] = gx * sc; p.g_diff[3 * (size_t)e + 1] = gy * sc; p.g_diff[3 * (size_t)e + 2] = gz * sc;
        const float gate = sigmoid_f(z + ba);
        s_gate[r] = gate;
        s_coef[r] = dot * gate * (1.0f - gate);
        s_dst[r] = i;
      }
    }
    __syncthreads();
    // phase 2: one thread per column, rows in sequence; column sums stay with the owning thread
    for (int c = tid; c < W; c += kThreads) {
      const float b = p.b2x[c], w = p.w3[c];
      float cb = 0.f, cw = 0.f;
      for (int r = 0; r < rows; ++r) {
        T* a = a2x + (size_t)(e0 + r) * W + c;
        float s, ds;
        silu_and_grad(Io<T>::ld(a) + b, s, ds);
        const float g = s_gsc[r] * w * ds;
        Io<T>::st(a, g);
        cb += g;
        cw = fmaf(s_gsc[r], s, cw);
      }
      cs[c] += cb; cs[W + c] += cw;
    }
    for (int c = tid; c < M; c += kThreads) {
      const float b = p.b2m[c], w = p.wa[c];
      float cb = 0.f, cw = 0.f;
      for (int r = 0; r < rows; ++r) {
        T* a = a2m + (size_t)(e0 + r) * M + c;
        float m, ds;
        silu_and_grad(Io<T>::ld(a) + b, m, ds);
        const float gm = fmaf(p.g_agg_m[(size_t)s_dst[r] * M + c], s_gate[r], s_coef[r] * w);
        const float g = gm * ds;
        Io<T>::st(a, g);
        cb += g;
        cw = fmaf(s_coef[r], m, cw);
      }
      cs[2 * W + c] += cb; cs[2 * W + M + c] += cw;
    }
    if (tid == 0)
      for (int r = 0; r < rows; ++r) { sum_gsc += s_gsc[r]; sum_coef += s_coef[r]; }
    __syncthreads();
  }
  for (int c = tid; c < W; c += kThreads) { atomicAdd(p.g_b2x + c, cs[c]); atomicAdd(p.g_w3 + c, cs[W + c]); }
  for (int c = tid; c < M; c += kThreads) { atomicAdd(p.g_b2m + c, cs[2 * W + c]); atomicAdd(p.g_wa + c, cs[2 * W + M + c]); }
  if (tid == 0) { atomicAdd(p.g_b3, sum_gsc); atomicAdd(p.g_ba, sum_coef); }
}

// ---- segment sums of the last edge pass, as dense per-node arrays --------------------------------------------
__global__ __launch_bounds__(kThreads) void agg_export_kernel(int N, int M, int MP, int R, int nsplit_x,
                                                               const int* __restrict__ row_ptr,
                                                               const float* __restrict__ agg_m, const float* __restrict__ part_m,
                                                               const float* __restrict__ agg_x, const float* __restrict__ part_x,
                                                               size_t agg_x_stride, size_t part_x_stride,
                                                               float* __restrict__ out_m, float* __restrict__ out_x) {
  const int n = blockIdx.x;
  const int rp0 = row_ptr[n], rp1 = row_ptr[n + 1];
  const bool any = rp1 > rp0;
  const int t0 = any ? rp0 / R : 0, t1 = any ? (rp1 - 1) / R : 0;
  for (int c = threadIdx.x; c < M; c += kThreads) {
    float v = 0.f;
    if (any) {
      if (t0 == t1) v = agg_m[(size_t)n * MP + c];
      else {
        v = part_m[((size_t)t0 * 2 + 1) * MP + c];
        for (int t = t0 + 1; t <= t1; ++t) v += part_m[((size_t)t * 2) * MP + c];
      }
    }
    out_m[(size_t)n * M + c] = v;
  }
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    float v = 0.f;
    if (any)
      for (int hs = 0; hs < nsplit_x; ++hs) {
        const float* ax = agg_x + (size_t)hs * agg_x_stride;
        const float* px = part_x + (size_t)hs * part_x_stride;
        if (t0 == t1) v += ax[(size_t)n * 4 + d];
        else {
          v += px[((size_t)t0 * 2 + 1) * 4 + d];
          for (int t = t0 + 1; t <= t1; ++t) v += px[((size_t)t * 2) * 4 + d];
        }
      }
    out_x[3 * (size_t)n + d] = v;
  }
}

}  // namespace
}  // namespace egnn

using namespace egnn;

extern "C" {

int egcl_read_aggregates(egnn_ctx* c, void* stream, int norm_scope, float* sum_m, float* sum_x, float* sq_sums) {
  if (!c || c->N == 0 || c->last_R <= 0) { set_error("egcl_read_aggregates: no layer has been run on this context"); return EGNN_ESTATE; }
  if (!sum_m || !sum_x || !sq_sums) { set_error("bad egcl_read_aggregates arguments"); return EGNN_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t agg_x_stride = (size_t)c->cap_nodes * 4, part_x_stride = (c->cap_tiles + 1) * 2 * 4;
  hipLaunchKernelGGL(agg_export_kernel, dim3(c->N), dim3(kThreads), 0, st, c->N, c->M, c->MP, c->last_R,
                     c->last_nsplit_x, c->row_ptr, c->agg_m, c->part_m, c->agg_x, c->part_x, agg_x_stride,
                     part_x_stride, sum_m, sum_x);
  EGNN_HIP(hipGetLastError());
  EGNN_HIP(hipMemcpyAsync(sq_sums, c->gscale, sizeof(float) * (norm_scope == EGNN_NORM_GRAPH ? c->B : 1),
                          hipMemcpyDeviceToDevice, st));
  return EGNN_OK;
}

static int bwd_l1(void* stream, int prec, int grad, int n_edges, int C, const int32_t* dst, const int32_t* src,
                  const float* P, const float* Q, const float* wd, const float* d2, void* buf) {
  if (n_edges < 0 || C <= 0 || (n_edges > 0 && (!dst || !src || !P || !Q || !wd || !d2 || !buf))) {
    set_error("bad egcl_backward first-layer arguments");
    return EGNN_EINVAL;
  }
  if (prec != EGNN_PREC_F32 && prec != EGNN_PREC_BF16) { set_error("bad precision %d", prec); return EGNN_EINVAL; }
  if (n_edges == 0) return EGNN_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((n_edges + kBwdRows - 1) / kBwdRows), block(kThreads);
  if (prec == EGNN_PREC_BF16) {
    if (grad) hipLaunchKernelGGL((bwd_l1_kernel<__bf16, true>), grid, block, 0, st, n_edges, C, dst, src, P, Q, wd, d2, static_cast<__bf16*>(buf));
    else hipLaunchKernelGGL((bwd_l1_kernel<__bf16, false>), grid, block, 0, st, n_edges, C, dst, src, P, Q, wd, d2, static_cast<__bf16*>(buf));
  } else {
    if (grad) hipLaunchKernelGGL((bwd_l1_kernel<float, true>), grid, block, 0, st, n_edges, C, dst, src, P, Q, wd, d2, static_cast<float*>(buf));
    else hipLaunchKernelGGL((bwd_l1_kernel<float, false>), grid, block, 0, st, n_edges, C, dst, src, P, Q, wd, d2, static_cast<float*>(buf));
  }
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egcl_backward_l1_act(void* stream, int prec, int n_edges, int C, const int32_t* dst, const int32_t* src,
                         const float* P, const float* Q, const float* wd, const float* d2, void* s1_out) {
  return bwd_l1(stream, prec, 0, n_edges, C, dst, src, P, Q, wd, d2, s1_out);
}

int egcl_backward_l1_grad(void* stream, int prec, int n_edges, int C, const int32_t* dst, const int32_t* src,
                          const float* P, const float* Q, const float* wd, const float* d2, void* g_s1_inout) {
  return bwd_l1(stream, prec, 1, n_edges, C, dst, src, P, Q, wd, d2, g_s1_inout);
}

int egcl_backward_heads(void* stream, int prec, int n_edges, int W, int M, const int32_t* dst, const int32_t* src,
                        const float* x, const float* g_sum_x, const float* g_sum_m, void* a2x_inout, void* a2m_inout,
                        const float* b2x, const float* w3, const float* b3, const float* b2m, const float* wa,
                        const float* ba, float* g_diff, float* g_b2x, float* g_w3, float* g_b3, float* g_b2m,
                        float* g_wa, float* g_ba) {
  if (n_edges < 0 || W <= 0 || M <= 0) { set_error("bad egcl_backward_heads sizes"); return EGNN_EINVAL; }
  if (prec != EGNN_PREC_F32 && prec != EGNN_PREC_BF16) { set_error("bad precision %d", prec); return EGNN_EINVAL; }
  if (n_edges == 0) return EGNN_OK;
  if (!dst || !src || !x || !g_sum_x || !g_sum_m || !a2x_inout || !a2m_inout || !b2x || !w3 || !b3 || !b2m || !wa ||
      !ba || !g_diff || !g_b2x || !g_w3 || !g_b3 || !g_b2m || !g_wa || !g_ba) {
    set_error("bad egcl_backward_heads arguments");
    return EGNN_EINVAL;
  }
  const size_t smem = (size_t)(2 * W + 2 * M) * sizeof(float);
  if (smem > 60 * 1024) { set_error("egcl_backward_heads: W + M = %d exceeds the column-sum buffer", W + M); return EGNN_EINVAL; }
  HeadsParams p;
  p.n_edges = n_edges; p.W = W; p.M = M; p.dst = dst; p.src = src; p.x = x; p.g_agg_x = g_sum_x; p.g_agg_m = g_sum_m;
  p.a2x = a2x_inout; p.a2m = a2m_inout; p.b2x = b2x; p.w3 = w3; p.b3 = b3; p.b2m = b2m; p.wa = wa; p.ba = ba;
  p.g_diff = g_diff; p.g_b2x = g_b2x; p.g_w3 = g_w3; p.g_b3 = g_b3; p.g_b2m = g_b2m; p.g_wa = g_wa; p.g_ba = g_ba;
  const int tiles = (n_edges + kBwdRows - 1) / kBwdRows;
  const dim3 grid(tiles < 2048 ? tiles : 2048), block(kThreads);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (prec == EGNN_PREC_BF16) hipLaunchKernelGGL(bwd_heads_kernel<__bf16>, grid, block, smem, st, p);
  else hipLaunchKernelGGL(bwd_heads_kernel<float>, grid, block, smem, st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // extern "C"
