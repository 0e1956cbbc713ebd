// The two small dense networks at the edge of the path, on the device without a BLAS library:
//   * gamma_tilde(t) of the learned noise schedule (SNR.py:50-52: l1(t) + l3(sigmoid(l2(l1(t)))) with softplus-positive
//     weights, PositiveLinear :5-22) for a grid of times -- what diffusion_x_h.py:40,46 evaluates over all T + 1 points;
//   * a ReLU MLP over node rows (SpectrumCompressor, DataPreprocessor.py:4-22: 200 -> 150 -> 100 -> 50 -> 32), evaluated
//     once per sample (the reference re-evaluates it every reverse step, parts/train_per_iretation.py:346).
// Both are tiny (T + 1 <= a few thousand points x 1024 hidden units; N x 51.6 k MAC) and off the per-step loop: plain
// fp32 kernels with fixed summation order (bitwise repeatable), no matrix cores.
#include <math.h>

#include "common.h"

namespace egnn {
namespace {

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch's threshold form

// one workgroup per time point: out[i] = sp(w1) t_i + sum_k sp(w3[k]) * sigmoid(sp(w2[k]) * sp(w1) t_i)
__global__ __launch_bounds__(256) void gamma_tilde_kernel(int hidden, const float* __restrict__ t, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, const float* __restrict__ w3,
                                                         float* __restrict__ out) {
  __shared__ float red[256];
  const float l1 = softplus_f(w1[0]) * t[blockIdx.x];
  float s = 0.f;
  for (int k = threadIdx.x; k < hidden; k += 256) s += softplus_f(w3[k]) * (1.0f / (1.0f + expf(-softplus_f(w2[k]) * l1)));
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = l1 + red[0];
}

// out[n][j] = act(b[j] + sum_k in[n][k] W[j][k]) for 16 rows per workgroup (inputs staged in LDS), nn.Linear layout W [J][K]
constexpr int kMlpRows = 16;
__global__ __launch_bounds__(256) void dense_rows_kernel(int N, int K, int J, const float* __restrict__ in, const float* __restrict__ W,
                                                        const float* __restrict__ b, int relu, float* __restrict__ out) {
  extern __shared__ float s_in[];   // [kMlpRows][K]
  const int n0 = blockIdx.x * kMlpRows, rows = min(kMlpRows, N - n0);
  for (int i = threadIdx.x; i < rows * K; i += 256) s_in[i] = in[(size_t)n0 * K + i];
  __syncthreads();
  for (int j = threadIdx.x; j < J; j += 256) {
    float acc[kMlpRows];
#pragma unroll
    for (int r = 0; r < kMlpRows; ++r) acc[r] = b[j];
    const float* wj = W + (size_t)j * K;
    for (int k = 0; k < K; ++k) {
      const float w = wj[k];
#pragma unroll
      for (int r = 0; r < kMlpRows; ++r) acc[r] = fmaf(s_in[r * K + k], w, acc[r]);   // rows beyond `rows` read stale LDS: never stored
    }
#pragma unroll
    for (int r = 0; r < kMlpRows; ++r)
      if (r < rows) out[(size_t)(n0 + r) * J + j] = relu ? fmaxf(acc[r], 0.f) : acc[r];
  }
}

}  // namespace

int launch_dense_rows(int N, int K, int J, const float* in, const float* W, const float* b, int relu, float* out, hipStream_t st) {
  // kMlpRows x K fp32 of dynamic LDS: 128 KiB at the documented limit K = 2048, above the 64 KiB a kernel gets by default
  static bool attr_done = false;
  if (!attr_done) {
    EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(dense_rows_kernel, dim3((N + kMlpRows - 1) / kMlpRows), dim3(256), (size_t)kMlpRows * K * sizeof(float), st, N, K,
                     J, in, W, b, relu, out);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}
int launch_gamma_tilde(int n, int hidden, const float* t, const float* w1, const float* w2, const float* w3, float* out, hipStream_t st) {
  hipLaunchKernelGGL(gamma_tilde_kernel, dim3(n), dim3(256), 0, st, hidden, t, w1, w2, w3, out);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace egnn

using namespace egnn;

extern "C" {

int egnn_gamma_tilde(void* stream, int n, int hidden, const float* d_t, const float* d_l1_w, const float* d_l2_w, const float* d_l3_w,
                     float* d_out) {
  if (n < 1 || hidden < 1 || !d_t || !d_l1_w || !d_l2_w || !d_l3_w || !d_out) { set_error("bad egnn_gamma_tilde arguments"); return EGNN_EINVAL; }
  return launch_gamma_tilde(n, hidden, d_t, d_l1_w, d_l2_w, d_l3_w, d_out, reinterpret_cast<hipStream_t>(stream));
}

int egnn_dense_rows(void* stream, int N, int K, int J, const float* d_in, const float* d_W, const float* d_b, int relu, float* d_out) {
  {
    const int rc = dense_rows_args_check(N, K, J, d_in, d_W, d_b, d_out);   // host_logic.cpp
    if (rc) return rc;
  }
  return launch_dense_rows(N, K, J, d_in, d_W, d_b, relu, d_out, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
