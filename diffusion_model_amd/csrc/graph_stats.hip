// Graph construction and evaluation statistics on the device.
//
//  * fully connected / radius graphs in the CSR layout the edge kernels consume -- counterpart of the
//    O(N^2) Python edge builders of the reference (parts/train_per_iretation.py:308-313,
//    make_dataset.py:131-136) and of PyG's collate offsets;
//  * RDF about atom 0 with Gaussian smoothing (evaluate_RDF.py:39-60) and the Si-O-Si selector / angle /
//    bond lengths (evaluate_Si-O-Si.py:23-53, CN2_evaluate.py:12-21), one workgroup per graph.
#include <math.h>

#include "common.h"

namespace egnn {

// ---- fully connected: node i of a graph with n atoms receives from every j != i, j ascending ----
__global__ void fc_rowptr_kernel(const int* __restrict__ graph_ptr, const int* __restrict__ node_graph, int N,
                                 const long long* __restrict__ edge_base, int* __restrict__ row_ptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > N) return;
  if (i == N) {
    const int g = node_graph[N - 1], lo = graph_ptr[g], n = graph_ptr[g + 1] - lo;
    row_ptr[N] = (int)(edge_base[g] + (long long)n * (n - 1));
    return;
  }
  const int g = node_graph[i], lo = graph_ptr[g], n = graph_ptr[g + 1] - lo;
  row_ptr[i] = (int)(edge_base[g] + (long long)(i - lo) * (n - 1));
}
__global__ void fc_fill_kernel(const int* __restrict__ graph_ptr, const int* __restrict__ node_graph, int N,
                               const int* __restrict__ row_ptr, int* __restrict__ edge_dst, int* __restrict__ edge_src) {
  // one workgroup per receiving node
  const int i = blockIdx.x;
  const int g = node_graph[i], lo = graph_ptr[g], n = graph_ptr[g + 1] - lo, base = row_ptr[i];
  for (int t = threadIdx.x; t < n - 1; t += blockDim.x) {
    const int j = lo + t + (lo + t >= i ? 1 : 0);
    edge_dst[base + t] = i;
    edge_src[base + t] = j;
  }
}

// ---- radius graph: j != i in the same graph with |x_i - x_j| < r, j ascending ----
__global__ void radius_count_kernel(const float* __restrict__ x, const int* __restrict__ graph_ptr,
                                    const int* __restrict__ node_graph, int N, float r2, int* __restrict__ deg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int g = node_graph[i], lo = graph_ptr[g], hi = graph_ptr[g + 1];
  const float a = x[3 * i], b = x[3 * i + 1], c = x[3 * i + 2];
  int cnt = 0;
  for (int j = lo; j < hi; ++j) {
    const float dx = a - x[3 * j], dy = b - x[3 * j + 1], dz = c - x[3 * j + 2];
    cnt += (j != i) && (dx * dx + dy * dy + dz * dz < r2);
  }
  deg[i] = cnt;
}
__global__ void radius_fill_kernel(const float* __restrict__ x, const int* __restrict__ graph_ptr,
                                   const int* __restrict__ node_graph, int N, float r2, const int* __restrict__ row_ptr,
                                   int* __restrict__ edge_dst, int* __restrict__ edge_src) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int g = node_graph[i], lo = graph_ptr[g], hi = graph_ptr[g + 1];
  const float a = x[3 * i], b = x[3 * i + 1], c = x[3 * i + 2];
  int e = row_ptr[i];
  for (int j = lo; j < hi; ++j) {
    const float dx = a - x[3 * j], dy = b - x[3 * j + 1], dz = c - x[3 * j + 2];
    if (j != i && dx * dx + dy * dy + dz * dz < r2) { edge_dst[e] = i; edge_src[e] = j; ++e; }
  }
}

// ---- RDF about atom 0 (evaluate_RDF.py:39-60) ----
// bins r_k = (k+1)*dR, k < nbins; count of distances with r_k < d < r_k + dR, divided by 4 pi rho r_k^2 dR with
// rho = n / (4/3 pi R^3); Gaussian filter (sigma bins, truncate 4 sigma, 'reflect' boundary) as scipy's
// gaussian_filter1d.  One workgroup per graph; fp64 accumulation like the numpy reference.
// Bin membership follows the reference's mixed arithmetic exactly (pinned by tests/golden/stats_golden.npz): the
// distance is a float32 torch.norm of a float32 difference, the bin edges are the float64 values
// np.arange(dR, R + dR, dR)[k] = dR + k*dR and r + dR, and `r < d < r + dR` compares them AFTER rounding the edges to
// float32 (a 0-dim float32 tensor against a Python float) -- so an atom sitting exactly on a float32 edge is in
// no bin.
constexpr int kMaxBins = 1024;
__global__ __launch_bounds__(256) void rdf_kernel(const float* __restrict__ pos, const int* __restrict__ graph_ptr, double R,
                                                  double dR, int nbins, float sigma, int normalize, float* __restrict__ out) {
  __shared__ int cnt[kMaxBins];
  __shared__ double raw[kMaxBins];
  __shared__ double red[256];
  const int g = blockIdx.x, lo = graph_ptr[g], hi = graph_ptr[g + 1], n = hi - lo;
  for (int k = threadIdx.x; k < nbins; k += blockDim.x) cnt[k] = 0;
  __syncthreads();
  const float x0 = pos[3 * lo], y0 = pos[3 * lo + 1], z0 = pos[3 * lo + 2];
  for (int i = lo + 1 + threadIdx.x; i < hi; i += blockDim.x) {
    const float dx = pos[3 * i] - x0, dy = pos[3 * i + 1] - y0, dz = pos[3 * i + 2] - z0;
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    const int k = (int)floor((double)d / dR) - 1;
    for (int kk = k - 1; kk <= k + 1; ++kk) {
      if (kk < 0 || kk >= nbins) continue;
      const double rk = dR + (double)kk * dR;
      if ((float)rk < d && d < (float)(rk + dR)) atomicAdd(&cnt[kk], 1);
    }
  }
  __syncthreads();
  const double rho = (double)n / (4.0 / 3.0 * M_PI * R * R * R);
  for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
    const double rk = dR + (double)k * dR;
    raw[k] = (double)cnt[k] / (4.0 * M_PI * rho * rk * rk * dR);
  }
  __syncthreads();
  const int lw = (int)(4.0 * (double)sigma + 0.5);
  double wsum = 0.0;
  for (int t = -lw; t <= lw; ++t) wsum += exp(-0.5 * (double)t * t / ((double)sigma * sigma));
  double mymax = 0.0;
  for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
    double acc = 0.0;
    for (int t = -lw; t <= lw; ++t) {
      int idx = k + t;
      const int period = 2 * nbins;   // 'reflect': (d c b a | a b c d | d c b a)
      idx %= period;
      if (idx < 0) idx += period;
      if (idx >= nbins) idx = period - 1 - idx;
      acc += raw[idx] * exp(-0.5 * (double)t * t / ((double)sigma * sigma));
    }
    acc /= wsum;
    out[(size_t)g * nbins + k] = (float)acc;
    mymax = fmax(mymax, acc);
  }
  if (normalize) {
    red[threadIdx.x] = mymax;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + w]);
      __syncthreads();
    }
    const double m = red[0];
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) out[(size_t)g * nbins + k] = (float)((double)out[(size_t)g * nbins + k] / m);
  }
}

// ---- Si-O-Si selector (evaluate_Si-O-Si.py:23-41) + CN2 angle / bond lengths (CN2_evaluate.py:12-21) ----
// per graph: atoms within `cutoff` of atom 0; valid iff exactly two and both one-hot [0,1] (Si).
// out[g] = {valid, angle_deg, len1, len2}
__global__ void sio_si_kernel(const float* __restrict__ pos, const int* __restrict__ onehot, int A,
                              const int* __restrict__ graph_ptr, int B, float cutoff, float* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B) return;
  const int lo = graph_ptr[g], hi = graph_ptr[g + 1];
  int idx[2] = {-1, -1}, cnt = 0;
  for (int i = lo + 1; i < hi; ++i) {
    const float dx = pos[3 * i] - pos[3 * lo], dy = pos[3 * i + 1] - pos[3 * lo + 1], dz = pos[3 * i + 2] - pos[3 * lo + 2];
    if (sqrtf(dx * dx + dy * dy + dz * dz) < cutoff) { if (cnt < 2) idx[cnt] = i; ++cnt; }
  }
  bool ok = cnt == 2;
  if (ok)
    for (int k = 0; k < 2; ++k) {
      bool si = A >= 2;
      for (int a = 0; a < A; ++a) si = si && (onehot[(size_t)idx[k] * A + a] == (a == 1 ? 1 : 0));
      ok = ok && si;
    }
  float ang = 0.f, l1 = 0.f, l2 = 0.f;
  if (ok) {
    float v1[3], v2[3];
    for (int d = 0; d < 3; ++d) { v1[d] = pos[3 * idx[0] + d] - pos[3 * lo + d]; v2[d] = pos[3 * idx[1] + d] - pos[3 * lo + d]; }
    l1 = sqrtf(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
    l2 = sqrtf(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]);
    const float c = (v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2]) / (l1 * l2);
    ang = acosf(c) * 57.29577951308232f;
  }
  out[4 * g] = ok ? 1.f : 0.f; out[4 * g + 1] = ang; out[4 * g + 2] = l1; out[4 * g + 3] = l2;
}

}  // namespace egnn

using namespace egnn;

extern "C" {

int egnn_fc_graph_build(void* stream, int N, int B, const int32_t* graph_ptr, const int32_t* node_graph,
                        const int64_t* edge_base, int32_t* row_ptr, int32_t* edge_dst, int32_t* edge_src) {
  if (N < 1 || B < 1 || !graph_ptr || !node_graph || !edge_base || !row_ptr) { set_error("bad egnn_fc_graph_build arguments"); return EGNN_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(fc_rowptr_kernel, dim3((N + 256) / 256), dim3(256), 0, st, graph_ptr, node_graph, N,
                     reinterpret_cast<const long long*>(edge_base), row_ptr);
  if (edge_dst && edge_src)
    hipLaunchKernelGGL(fc_fill_kernel, dim3(N), dim3(64), 0, st, graph_ptr, node_graph, N, row_ptr, edge_dst, edge_src);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egnn_radius_graph_count(void* stream, int N, const float* x, const int32_t* graph_ptr, const int32_t* node_graph,
                            float radius, int32_t* deg) {
  if (N < 1 || !x || !graph_ptr || !node_graph || !deg || !(radius > 0.f)) { set_error("bad egnn_radius_graph_count arguments"); return EGNN_EINVAL; }
  hipLaunchKernelGGL(radius_count_kernel, dim3((N + 127) / 128), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), x,
                     graph_ptr, node_graph, N, radius * radius, deg);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egnn_radius_graph_fill(void* stream, int N, const float* x, const int32_t* graph_ptr, const int32_t* node_graph,
                           float radius, const int32_t* row_ptr, int32_t* edge_dst, int32_t* edge_src) {
  if (N < 1 || !x || !graph_ptr || !node_graph || !row_ptr || !edge_dst || !edge_src) { set_error("bad egnn_radius_graph_fill arguments"); return EGNN_EINVAL; }
  hipLaunchKernelGGL(radius_fill_kernel, dim3((N + 127) / 128), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), x,
                     graph_ptr, node_graph, N, radius * radius, row_ptr, edge_dst, edge_src);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egnn_rdf(void* stream, int B, const float* pos, const int32_t* graph_ptr, double R, double dR, float sigma,
             int normalize, int nbins, float* out) {
  if (B < 1 || !pos || !graph_ptr || !out || nbins < 1 || nbins > kMaxBins || !(dR > 0.0) || !(sigma > 0.f)) {
    set_error("bad egnn_rdf arguments (nbins <= %d)", kMaxBins);
    return EGNN_EINVAL;
  }
  hipLaunchKernelGGL(rdf_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pos, graph_ptr, R, dR,
                     nbins, sigma, normalize, out);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egnn_si_o_si(void* stream, int B, int A, const float* pos, const int32_t* onehot, const int32_t* graph_ptr,
                 float cutoff, float* out) {
  if (B < 1 || A < 1 || !pos || !onehot || !graph_ptr || !out) { set_error("bad egnn_si_o_si arguments"); return EGNN_EINVAL; }
  hipLaunchKernelGGL(sio_si_kernel, dim3((B + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), pos, onehot, A,
                     graph_ptr, B, cutoff, out);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // extern "C"
