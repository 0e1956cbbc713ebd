// Backward stage kernels of one EGCL layer (training, reference EquivariantGraphNeuralNetwork.py:55-71 under
// autograd).  The backward of a layer is a chain
//     first-layer activations -> [GEMM] -> heads -> [GEMM x2] -> first-layer gradient -> [GEMM x2]
// in which the bracketed steps are plain dense GEMMs (dgrad / wgrad of mlp_x.2, mlp_m.2 and of the two first
// Linear layers) that the host runs with the BLAS library, and everything between them -- gathers, SiLU and its
// derivative, the scalar heads (mlp_x.4, the attention gate), row reductions and the column-sum gradients of
// biases / w3 / wa -- is fused into the three kernels below, one HBM pass each, in place on the GEMM buffers.
// Storage type of the [edges, W] buffers follows the precision (float or bf16); all arithmetic is fp32.
#include <stdlib.h>

#include "bwd_graph.h"
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

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
// V consecutive elements (V = 1 or 4; 4 = one 16-byte float or 8-byte bf16 access)
template <typename T, int V> struct VecIo;
template <typename T> struct VecIo<T, 1> {
  static __device__ __forceinline__ void ld(const T* p, float (&o)[1]) { o[0] = Io<T>::ld(p); }
  static __device__ __forceinline__ void st(T* p, const float (&v)[1]) { Io<T>::st(p, v[0]); }
};
template <> struct VecIo<float, 4> {
  static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j];
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j];
    *reinterpret_cast<f32x4*>(p) = o;
  }
};
template <> struct VecIo<__bf16, 4> {
  static __device__ __forceinline__ void ld(const __bf16* p, float (&o)[4]) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (float)v[j];
  }
  static __device__ __forceinline__ void st(__bf16* p, const float (&v)[4]) {
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x4*>(p) = o;
  }
};

template <> struct VecIo<float, 8> {
  static __device__ __forceinline__ void ld(const float* p, float (&o)[8]) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = v0[j]; o[j + 4] = v1[j]; }
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[8]) {
    f32x4 o0, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o0[j] = v[j]; o1[j] = v[j + 4]; }
    *reinterpret_cast<f32x4*>(p) = o0;
    *reinterpret_cast<f32x4*>(p + 4) = o1;
  }
};
template <> struct VecIo<__bf16, 8> {
  static __device__ __forceinline__ void ld(const __bf16* p, float (&o)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
  }
  static __device__ __forceinline__ void st(__bf16* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8*>(p) = o;
  }
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
// C = columns of the table.  GRAD = false writes the activation, GRAD = true multiplies the buffer in place by
// SiLU'(pre-activation)  (dL/da1 = dL/ds1 * SiLU'(a1)).  Work items = (row, group of V columns), flat over the
// tile so that any C keeps all threads busy.
template <typename T, bool GRAD, int V>
__global__ __launch_bounds__(kThreads) void bwd_l1_kernel(int n_edges, int C, const int* __restrict__ dst,
                                                           const int* __restrict__ src, const float* __restrict__ P,
                                                           const float* __restrict__ Q, const float* __restrict__ wd,
                                                           const float* __restrict__ d2, T* __restrict__ buf) {
  const int e0 = blockIdx.x * kBwdRows, rows = min(kBwdRows, n_edges - e0);
  const int G = C / V, items = rows * G;
#pragma unroll 4
  for (int it = threadIdx.x; it < items; it += kThreads) {
    const int r = it / G, c = (it - r * G) * V, e = e0 + r;
    float p[V], q[V], w[V], o[V];
    VecIo<float, V>::ld(P + (size_t)dst[e] * C + c, p);
    VecIo<float, V>::ld(Q + (size_t)src[e] * C + c, q);
    VecIo<float, V>::ld(wd + c, w);
    const float d = d2[e];
    T* optr = buf + (size_t)e * C + c;
    if (GRAD) VecIo<T, V>::ld(optr, o);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float sv, ds;
      silu_and_grad(fmaf(w[j], d, p[j] + q[j]), sv, ds);
      o[j] = GRAD ? o[j] * ds : sv;
    }
    VecIo<T, V>::st(optr, o);
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

template <typename T, int V>
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
      for (int c = V * lane; c < W; c += V * 64) {
        float a[V], b[V], w[V];
        VecIo<T, V>::ld(a2x + (size_t)e * W + c, a);
        VecIo<float, V>::ld(p.b2x + c, b);
        VecIo<float, V>::ld(p.w3 + c, w);
#pragma unroll
        for (int k = 0; k < V; ++k) acc = fmaf(w[k], silu_f(a[k] + b[k]), acc);
      }
      float z = 0.f, dot = 0.f;
      for (int c = V * lane; c < M; c += V * 64) {
        float a[V], b[V], w[V], g[V];
        VecIo<T, V>::ld(a2m + (size_t)e * M + c, a);
        VecIo<float, V>::ld(p.b2m + c, b);
        VecIo<float, V>::ld(p.wa + c, w);
        VecIo<float, V>::ld(p.g_agg_m + (size_t)i * M + c, g);
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const float m = silu_f(a[k] + b[k]);
          z = fmaf(w[k], m, z);
          dot = fmaf(g[k], m, dot);
        }
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
    // phase 2: one thread per group of V columns, rows in sequence; column sums stay with the owning thread
    for (int c = V * tid; c < W; c += V * kThreads) {
      float b[V], w[V], cb[V], cw[V];
      VecIo<float, V>::ld(p.b2x + c, b);
      VecIo<float, V>::ld(p.w3 + c, w);
#pragma unroll
      for (int k = 0; k < V; ++k) cb[k] = cw[k] = 0.f;
#pragma unroll 4
      for (int r = 0; r < rows; ++r) {
        T* ap = a2x + (size_t)(e0 + r) * W + c;
        float a[V];
        VecIo<T, V>::ld(ap, a);
        const float gsc = s_gsc[r];
#pragma unroll
        for (int k = 0; k < V; ++k) {
          float sv, ds;
          silu_and_grad(a[k] + b[k], sv, ds);
          a[k] = gsc * w[k] * ds;
          cb[k] += a[k];
          cw[k] = fmaf(gsc, sv, cw[k]);
        }
        VecIo<T, V>::st(ap, a);
      }
#pragma unroll
      for (int k = 0; k < V; ++k) { cs[c + k] += cb[k]; cs[W + c + k] += cw[k]; }
    }
    for (int c = V * tid; c < M; c += V * kThreads) {
      float b[V], w[V], cb[V], cw[V];
      VecIo<float, V>::ld(p.b2m + c, b);
      VecIo<float, V>::ld(p.wa + c, w);
#pragma unroll
      for (int k = 0; k < V; ++k) cb[k] = cw[k] = 0.f;
#pragma unroll 4
      for (int r = 0; r < rows; ++r) {
        T* ap = a2m + (size_t)(e0 + r) * M + c;
        float a[V], g[V];
        VecIo<T, V>::ld(ap, a);
        VecIo<float, V>::ld(p.g_agg_m + (size_t)s_dst[r] * M + c, g);
        const float gate = s_gate[r], coef = s_coef[r];
#pragma unroll
        for (int k = 0; k < V; ++k) {
          float m, ds;
          silu_and_grad(a[k] + b[k], m, ds);
          a[k] = fmaf(g[k], gate, coef * w[k]) * ds;
          cb[k] += a[k];
          cw[k] = fmaf(coef, m, cw[k]);
        }
        VecIo<T, V>::st(ap, a);
      }
#pragma unroll
      for (int k = 0; k < V; ++k) { cs[2 * W + c + k] += cb[k]; cs[2 * W + M + c + k] += cw[k]; }
    }
    if (tid == 0)
      for (int r = 0; r < rows; ++r) { sum_gsc += s_gsc[r]; sum_coef += s_coef[r]; }
    __syncthreads();
  }
  for (int c = tid; c < W; c += kThreads) { atomicAdd(p.g_b2x + c, cs[c]); atomicAdd(p.g_w3 + c, cs[W + c]); }
  for (int c = tid; c < M; c += kThreads) { atomicAdd(p.g_b2m + c, cs[2 * W + c]); atomicAdd(p.g_wa + c, cs[2 * W + M + c]); }
  if (tid == 0) { atomicAdd(p.g_b3, sum_gsc); atomicAdd(p.g_ba, sum_coef); }
}

// ---- gather of the first Linear layers' input: in[e] = [h_i | h_j | d2 | 1 | 0...] (:56) ------------------------
// K1P >= 2H + 2 columns (the ones column carries the bias gradient through the wgrad GEMM); also writes d2[e].
template <typename T>
__global__ __launch_bounds__(kThreads) void bwd_gather_kernel(int n_edges, int H, int K1P, const int* __restrict__ dst,
                                                               const int* __restrict__ src, const float* __restrict__ h,
                                                               const float* __restrict__ x, T* __restrict__ inp,
                                                               float* __restrict__ d2) {
  const size_t total = (size_t)n_edges * K1P;
  for (size_t t = (size_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += (size_t)gridDim.x * kThreads) {
    const int e = (int)(t / K1P), c = (int)(t - (size_t)e * K1P);
    float v = 0.f;
    if (c < H) v = h[(size_t)dst[e] * H + c];
    else if (c < 2 * H) v = h[(size_t)src[e] * H + c - H];
    else if (c == 2 * H) {
      const int i = dst[e], j = src[e];
      const float dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
      v = dx * dx + dy * dy + dz * dz;
      d2[e] = v;
    } else if (c == 2 * H + 1) v = 1.f;
    Io<T>::st(inp + t, v);
  }
}

// bf16 rows with K1P % 8 == 0: one 16-byte piece (8 columns) per thread, i.e. K1P / 8 neighbouring lanes write one contiguous
// row; the element-per-thread form above ran the [E, 128] bf16 operand at 0.7 TB/s (0.36 ms per layer at C4 shapes).
__global__ __launch_bounds__(kThreads) void bwd_gather_rows_kernel(int n_edges, int H, int K1P, const int* __restrict__ dst,
                                                                    const int* __restrict__ src, const float* __restrict__ h,
                                                                    const float* __restrict__ x, __bf16* __restrict__ inp,
                                                                    float* __restrict__ d2) {
  const int pieces = K1P >> 3;
  const size_t total = (size_t)n_edges * pieces;
  for (size_t t = (size_t)blockIdx.x * kThreads + threadIdx.x; t < total; t += (size_t)gridDim.x * kThreads) {
    const int e = (int)(t / pieces), c0 = 8 * (int)(t - (size_t)e * pieces);
    const int i = dst[e], j = src[e];
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = c0 + k;
      float v = 0.f;
      if (c < H) v = h[(size_t)i * H + c];
      else if (c < 2 * H) v = h[(size_t)j * H + c - H];
      else if (c == 2 * H) {
        const float dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
        v = dx * dx + dy * dy + dz * dz;
        d2[e] = v;
      } else if (c == 2 * H + 1) v = 1.f;
      o[k] = (__bf16)v;
    }
    *reinterpret_cast<bf16x8*>(inp + (size_t)e * K1P + c0) = o;
  }
}

// ---- scatter of dL/d(in) back to the nodes ------------------------------------------------------------------------
// g_in[e] = dL/d[h_i | h_j | d2] (K1P columns, the rest ignored).  g_h[i] += g_in[:H], g_h[j] += g_in[H:2H];
// dL/d(x_i - x_j) = g_diff[e] + 2 (g_in[2H] + g_S[segment of i]) (x_i - x_j)  goes to g_x[i] and, negated, g_x[j].
// Edges arrive sorted by receiving node, so the dst side is summed over runs of equal dst in registers and added
// once per run; the src side is one atomic add per edge.  One workgroup = kScatRows consecutive edges, one thread
// per column of [g_h (dst) | g_h (src) | g_x].
constexpr int kScatRows = 64;
// ---- node MLP backward, activation stage (mlp_h: Linear -> SiLU -> Linear, EquivariantGraphNeuralNetwork.py:26-30 under autograd):
//     s = SiLU(z + b1) and dL/dz = dL/ds * SiLU'(z + b1) as the bf16 operands of the weight-gradient / dgrad products, and the
//     bias gradient (column sums of dL/dz) -- one pass instead of a dozen element-wise launches over [N, W] fp32.
// block = 256 threads = 4 row lanes x 64 column quads (256 columns), 64 rows per block
__global__ __launch_bounds__(256) void node_act_bwd_kernel(int N, int W, const float* __restrict__ z, int ldz, const float* __restrict__ b1,
                                                           const float* __restrict__ gs, int ldg, __bf16* __restrict__ gz_out,
                                                           __bf16* __restrict__ s_out, int ldo, float* __restrict__ g_b1) {
  __shared__ float red[4][256];
  const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = 256 * blockIdx.y + 4 * cq;
  const int r0 = 64 * blockIdx.x;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < W) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(b1 + c);
    for (int i = 0; i < 16; ++i) {
      const int row = r0 + rl + 4 * i;
      if (row >= N) break;
      const f32x4 zv = *reinterpret_cast<const f32x4*>(z + (size_t)row * ldz + c);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gs + (size_t)row * ldg + c);
      __bf16 go[4], so[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = zv[k] + bb[k];
        const float sg = 1.0f / (1.0f + __expf(-a));
        const float g = gv[k] * (sg * (1.0f + a * (1.0f - sg)));
        cs[k] += g;
        go[k] = (__bf16)g;
        so[k] = (__bf16)(a * sg);
      }
      *reinterpret_cast<uint2*>(gz_out + (size_t)row * ldo + c) = *reinterpret_cast<const uint2*>(go);
      *reinterpret_cast<uint2*>(s_out + (size_t)row * ldo + c) = *reinterpret_cast<const uint2*>(so);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[rl][4 * cq + k] = cs[k];
  __syncthreads();
  const int cc = 256 * blockIdx.y + threadIdx.x;
  if (cc < W) atomicAdd(g_b1 + cc, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bwd_scatter_kernel(int n_edges, int H, int K1P, const int* __restrict__ dst,
                                                                const int* __restrict__ src, const float* __restrict__ x,
                                                                const T* __restrict__ g_in, const float* __restrict__ g_diff,
                                                                const float* __restrict__ g_S, const int* __restrict__ node_seg,
                                                                float* __restrict__ g_h, float* __restrict__ g_x) {
  const int e0 = blockIdx.x * kScatRows, e1 = min(e0 + kScatRows, n_edges);
  for (int c = threadIdx.x; c < 2 * H + 3; c += kThreads) {
    if (c >= H && c < 2 * H) {   // sending side: scattered
      for (int e = e0; e < e1; ++e) atomicAdd(g_h + (size_t)src[e] * H + c - H, Io<T>::ld(g_in + (size_t)e * K1P + c));
      continue;
    }
    const int d = c - 2 * H;     // coordinate component when c >= 2H
    float run = 0.f;
    int cur = dst[e0];
    for (int e = e0; e < e1; ++e) {
      const int i = dst[e];
      if (i != cur) {
        atomicAdd(c < H ? g_h + (size_t)cur * H + c : g_x + 3 * (size_t)cur + d, run);
        run = 0.f; cur = i;
      }
      if (c < H) run += Io<T>::ld(g_in + (size_t)e * K1P + c);
      else {
        const int j = src[e];
        const float gd2 = Io<T>::ld(g_in + (size_t)e * K1P + 2 * H) + g_S[node_seg ? node_seg[i] : 0];
        const float g = fmaf(2.0f * gd2, x[3 * i + d] - x[3 * j + d], g_diff[3 * (size_t)e + d]);
        run += g;
        atomicAdd(g_x + 3 * (size_t)j + d, -g);
      }
    }
    atomicAdd(c < H ? g_h + (size_t)cur * H + c : g_x + 3 * (size_t)cur + d, run);
  }
}

// g_diff[e] = dL/d(sum_x[i]) * s_e with s_e = sum of the column-split shares the coordinate recompute kernels wrote
__global__ void bwd_gdiff_kernel(int n_edges, int nsplit, size_t share_stride, const int* __restrict__ dst,
                                 const float* __restrict__ g_sum_x, const float* __restrict__ s_halves,
                                 float* __restrict__ g_diff) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  float sv = 0.f;
  for (int hs = 0; hs < nsplit; ++hs) sv += s_halves[(size_t)hs * share_stride + e];
  const int i = dst[e];
  g_diff[3 * (size_t)e] = g_sum_x[3 * i] * sv;
  g_diff[3 * (size_t)e + 1] = g_sum_x[3 * i + 1] * sv;
  g_diff[3 * (size_t)e + 2] = g_sum_x[3 * i + 2] * sv;
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

int egcl_backward_fused_supported(egnn_ctx* c) {
  if (!c || c->L == 0 || c->N == 0) return 0;
  return backward_recompute_supported(c);
}

int egcl_backward_table(egnn_ctx* c, void* stream, int layer, const float* h) {
  if (!c || c->L == 0 || c->N == 0 || layer < 0 || layer >= c->L || !h) { set_error("bad egcl_backward_table arguments"); return EGNN_EINVAL; }
  if (!c->layers[layer].packed) { set_error("layer %d has no packed parameters", layer); return EGNN_ESTATE; }
  if (!backward_recompute_supported(c)) { set_error("fused backward recompute is not available for these widths"); return EGNN_EINVAL; }
  return backward_table(c, reinterpret_cast<hipStream_t>(stream), layer, h);
}

int egcl_backward_edge_recompute(egnn_ctx* c, void* stream, int layer, const float* x, const float* g_sum_x,
                                 const float* g_sum_m, int e_first, int n_edges, void* s1x, void* s1m, void* g_a2x,
                                 void* g_a2m, float* g_diff, float* g_b2x, float* g_w3, float* g_b3, float* g_b2m,
                                 float* g_wa, float* g_ba) {
  if (!c || c->L == 0 || c->N == 0 || layer < 0 || layer >= c->L) { set_error("bad egcl_backward_edge_recompute context/layer"); return EGNN_EINVAL; }
  if (e_first < 0 || n_edges < 0 || e_first + n_edges > c->E) { set_error("edge range [%d, %d) outside the graph", e_first, e_first + n_edges); return EGNN_EINVAL; }
  if (n_edges == 0) return EGNN_OK;
  if (!x || !g_sum_x || !g_sum_m || !s1x || !s1m || !g_a2x || !g_a2m || !g_diff || !g_b2x || !g_w3 || !g_b3 || !g_b2m ||
      !g_wa || !g_ba) { set_error("bad egcl_backward_edge_recompute arguments"); return EGNN_EINVAL; }
  if (!backward_recompute_supported(c)) { set_error("fused backward recompute is not available for these widths"); return EGNN_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nsplit = c->WxP >= 512 ? c->WxP / 512 : 1;
  if ((size_t)nsplit * n_edges > c->cap_bwd_s) {
    if (c->bwd_s) { (void)hipFree(c->bwd_s); c->bwd_s = nullptr; c->cap_bwd_s = 0; }
    EGNN_HIP(hipMalloc(reinterpret_cast<void**>(&c->bwd_s), (size_t)nsplit * n_edges * sizeof(float)));
    c->cap_bwd_s = (size_t)nsplit * n_edges;
    const char* poison = getenv("EGNN_DEBUG_POISON");   // (tests: see dev_alloc in egnn_forward.hip)
    if (poison && poison[0] == '1') EGNN_HIP(hipMemset(c->bwd_s, 0xFF, c->cap_bwd_s * sizeof(float)));
  }
  int rc = backward_recompute(c, st, layer, x, g_sum_x, g_sum_m, e_first, n_edges, s1x, s1m, g_a2x, g_a2m, c->bwd_s, g_b2x,
                              g_w3, g_b3, g_b2m, g_wa, g_ba);
  if (rc) return rc;
  hipLaunchKernelGGL(bwd_gdiff_kernel, dim3((n_edges + 255) / 256), dim3(256), 0, st, n_edges, nsplit, (size_t)n_edges,
                     c->edge_dst + e_first, g_sum_x, c->bwd_s, g_diff);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egcl_backward_heads_saved(egnn_ctx* c, void* stream, int layer, const float* x, const float* g_sum_x, const float* g_sum_m,
                              int e_first, int n_edges, void* t2x, void* t2m, const float* s_shares, float* g_diff,
                              float* g_b2x, float* g_w3, float* g_b3, float* g_b2m, float* g_wa, float* g_ba) {
  if (!c || c->L == 0 || c->N == 0 || layer < 0 || layer >= c->L) { set_error("bad egcl_backward_heads_saved context/layer"); return EGNN_EINVAL; }
  if (e_first < 0 || n_edges < 0 || e_first + n_edges > c->E) { set_error("edge range [%d, %d) outside the graph", e_first, e_first + n_edges); return EGNN_EINVAL; }
  if (n_edges == 0) return EGNN_OK;
  if (!x || !g_sum_x || !g_sum_m || !t2x || !t2m || !s_shares || !g_diff || !g_b2x || !g_w3 || !g_b3 || !g_b2m || !g_wa ||
      !g_ba) { set_error("bad egcl_backward_heads_saved arguments"); return EGNN_EINVAL; }
  if (!backward_recompute_supported(c)) { set_error("the saved-activation backward is not available for these widths"); return EGNN_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc = backward_heads_saved(c, st, layer, x, g_sum_x, g_sum_m, e_first, n_edges, t2x, t2m, g_b2x, g_w3, g_b3, g_b2m, g_wa, g_ba);
  if (rc) return rc;
  const int nsplit = c->WxP >= 512 ? c->WxP / 512 : 1;
  // s_shares: [nsplit][E] shares of s_e of the WHOLE layer as egcl_forward_save left them
  hipLaunchKernelGGL(bwd_gdiff_kernel, dim3((n_edges + 255) / 256), dim3(256), 0, st, n_edges, nsplit, (size_t)c->E,
                     c->edge_dst + e_first, g_sum_x, s_shares + e_first, g_diff);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egcl_backward_dgrad(egnn_ctx* c, void* stream, int layer, const float* x, int e_first, int n_edges, const void* g_a2x,
                        const void* g_a2m, void* g_a1x, void* g_a1m) {
  if (!c || c->L == 0 || c->N == 0 || layer < 0 || layer >= c->L) { set_error("bad egcl_backward_dgrad context/layer"); return EGNN_EINVAL; }
  if (e_first < 0 || n_edges < 0 || e_first + n_edges > c->E) { set_error("edge range [%d, %d) outside the graph", e_first, e_first + n_edges); return EGNN_EINVAL; }
  if (n_edges == 0) return EGNN_OK;
  if (!x || !g_a2x || !g_a2m || !g_a1x || !g_a1m) { set_error("bad egcl_backward_dgrad arguments"); return EGNN_EINVAL; }
  if (!backward_recompute_supported(c)) { set_error("fused backward is not available for these widths"); return EGNN_EINVAL; }
  return backward_dgrad(c, reinterpret_cast<hipStream_t>(stream), layer, x, e_first, n_edges, g_a2x, g_a2m, g_a1x, g_a1m);
}

int egcl_backward_dgrad_reduce(egnn_ctx* c, void* stream, int layer, const float* x, int e_first, int n_edges, const void* g_a2x,
                               const void* g_a2m, void* G, float* cd_x, float* cd_m, float* gd2_part) {
  if (!c || c->L == 0 || c->N == 0 || layer < 0 || layer >= c->L) { set_error("bad egcl_backward_dgrad_reduce context/layer"); return EGNN_EINVAL; }
  if (e_first < 0 || n_edges < 0 || e_first + n_edges > c->E) { set_error("edge range [%d, %d) outside the graph", e_first, e_first + n_edges); return EGNN_EINVAL; }
  if (n_edges == 0) return EGNN_OK;
  if (!x || !g_a2x || !g_a2m || !G || !cd_x || !cd_m || !gd2_part) { set_error("bad egcl_backward_dgrad_reduce arguments"); return EGNN_EINVAL; }
  if (!backward_recompute_supported(c)) { set_error("fused backward is not available for these widths"); return EGNN_EINVAL; }
  if (!c->graph_ptr || !c->row_ptr || c->B < 1) { set_error("egcl_backward_dgrad_reduce needs the graph ranges (egnn_set_graph)"); return EGNN_ESTATE; }
  return backward_dgrad_graph(c, reinterpret_cast<hipStream_t>(stream), layer, x, e_first, n_edges, g_a2x, g_a2m, G, cd_x, cd_m,
                              gd2_part);
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
  const bool v4 = (C & 3) == 0, v8 = (C & 7) == 0;
#define L1_LAUNCH(T, G, V) \
  hipLaunchKernelGGL((bwd_l1_kernel<T, G, V>), grid, block, 0, st, n_edges, C, dst, src, P, Q, wd, d2, static_cast<T*>(buf))
  if (prec == EGNN_PREC_BF16) {
    if (grad) { if (v8) L1_LAUNCH(__bf16, true, 8); else if (v4) L1_LAUNCH(__bf16, true, 4); else L1_LAUNCH(__bf16, true, 1); }
    else { if (v8) L1_LAUNCH(__bf16, false, 8); else if (v4) L1_LAUNCH(__bf16, false, 4); else L1_LAUNCH(__bf16, false, 1); }
  } else {
    if (grad) { if (v4) L1_LAUNCH(float, true, 4); else L1_LAUNCH(float, true, 1); }
    else { if (v4) L1_LAUNCH(float, false, 4); else L1_LAUNCH(float, false, 1); }
  }
#undef L1_LAUNCH
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
  const bool v4 = (W & 3) == 0 && (M & 3) == 0;
  if (prec == EGNN_PREC_BF16) {
    if (v4) hipLaunchKernelGGL((bwd_heads_kernel<__bf16, 4>), grid, block, smem, st, p);
    else hipLaunchKernelGGL((bwd_heads_kernel<__bf16, 1>), grid, block, smem, st, p);
  } else {
    if (v4) hipLaunchKernelGGL((bwd_heads_kernel<float, 4>), grid, block, smem, st, p);
    else hipLaunchKernelGGL((bwd_heads_kernel<float, 1>), grid, block, smem, st, p);
  }
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egcl_backward_gather_in(void* stream, int prec, int n_edges, int H, int K1P, const int32_t* dst, const int32_t* src,
                            const float* h, const float* x, void* in_out, float* d2_out) {
  if (n_edges < 0 || H <= 0 || K1P < 2 * H + 2) { set_error("bad egcl_backward_gather_in sizes"); return EGNN_EINVAL; }
  if (prec != EGNN_PREC_F32 && prec != EGNN_PREC_BF16) { set_error("bad precision %d", prec); return EGNN_EINVAL; }
  if (n_edges == 0) return EGNN_OK;
  if (!dst || !src || !h || !x || !in_out || !d2_out) { set_error("bad egcl_backward_gather_in arguments"); return EGNN_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)n_edges * K1P;
  const dim3 grid((unsigned)((total + kThreads - 1) / kThreads < 65536 ? (total + kThreads - 1) / kThreads : 65536)), block(kThreads);
  if (prec == EGNN_PREC_BF16 && K1P % 8 == 0) {
    const size_t pieces = (size_t)n_edges * (K1P / 8);
    const dim3 grid8((unsigned)((pieces + kThreads - 1) / kThreads < 65536 ? (pieces + kThreads - 1) / kThreads : 65536));
    hipLaunchKernelGGL(bwd_gather_rows_kernel, grid8, block, 0, st, n_edges, H, K1P, dst, src, h, x, static_cast<__bf16*>(in_out), d2_out);
  } else if (prec == EGNN_PREC_BF16)
    hipLaunchKernelGGL(bwd_gather_kernel<__bf16>, grid, block, 0, st, n_edges, H, K1P, dst, src, h, x, static_cast<__bf16*>(in_out), d2_out);
  else
    hipLaunchKernelGGL(bwd_gather_kernel<float>, grid, block, 0, st, n_edges, H, K1P, dst, src, h, x, static_cast<float*>(in_out), d2_out);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egcl_backward_node_act(void* stream, int N, int W, const float* z, int ldz, const float* b1, const float* g_s, int ldg,
                           void* g_z_out, void* s_out, int ldo, float* g_b1) {
  if (N < 1 || W < 4 || W % 4 != 0 || ldz < W || ldg < W || ldo < W || ldz % 4 != 0 || ldg % 4 != 0 || ldo % 4 != 0 || !z || !b1 || !g_s ||
      !g_z_out || !s_out || !g_b1) {
    set_error("bad egcl_backward_node_act arguments (W and the row strides multiples of 4)");
    return EGNN_EINVAL;
  }
  hipLaunchKernelGGL(node_act_bwd_kernel, dim3((N + 63) / 64, (W + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), N, W, z,
                     ldz, b1, g_s, ldg, static_cast<__bf16*>(g_z_out), static_cast<__bf16*>(s_out), ldo, g_b1);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egcl_backward_scatter(void* stream, int prec, int n_edges, int H, int K1P, const int32_t* dst, const int32_t* src,
                          const float* x, const void* g_in, const float* g_diff, const float* g_sq_sums,
                          const int32_t* node_segment, float* g_h, float* g_x) {
  if (n_edges < 0 || H <= 0 || K1P < 2 * H + 1) { set_error("bad egcl_backward_scatter sizes"); return EGNN_EINVAL; }
  if (prec != EGNN_PREC_F32 && prec != EGNN_PREC_BF16) { set_error("bad precision %d", prec); return EGNN_EINVAL; }
  if (n_edges == 0) return EGNN_OK;
  if (!dst || !src || !x || !g_in || !g_diff || !g_sq_sums || !g_h || !g_x) { set_error("bad egcl_backward_scatter arguments"); return EGNN_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((n_edges + kScatRows - 1) / kScatRows), block(2 * H + 3 <= 128 ? 128 : kThreads);
  if (prec == EGNN_PREC_BF16)
    hipLaunchKernelGGL(bwd_scatter_kernel<__bf16>, grid, block, 0, st, n_edges, H, K1P, dst, src, x, static_cast<const __bf16*>(g_in), g_diff, g_sq_sums, node_segment, g_h, g_x);
  else
    hipLaunchKernelGGL(bwd_scatter_kernel<float>, grid, block, 0, st, n_edges, H, K1P, dst, src, x, static_cast<const float*>(g_in), g_diff, g_sq_sums, node_segment, g_h, g_x);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // extern "C"
