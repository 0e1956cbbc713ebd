// Diffusion-process arithmetic and the device-resident reverse (p_sample) loop for gfx950.
//
// Reference: diffusion_x_h.py (schedule :92-106, calculate_mu :61-73, reverse_diffuse_one_step
// :75-90, remove_mean :5-14) and the generate() loop of parts/train_per_iretation.py:264-444.
// The reference performs ~50 tiny launches and >= 3 host syncs per reverse step; here one reverse
// step is the EGNN forward plus ONE fused per-graph kernel (epsilon extraction, remove_mean, mu,
// noise, state update, non-finite flag), the step index lives in device memory and the whole step
// is replayed from a hipGraph.
#include <math.h>

#include "common.h"

namespace egnn {

// ---- Philox4x32-10 counter-based generator -------------------------------------------------------
struct u4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u4 philox4x32(u4 ctr, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * ctr.x, p1 = (uint64_t)0xCD9E8D57u * ctr.z;
    u4 n;
    n.x = (uint32_t)(p1 >> 32) ^ ctr.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ ctr.w ^ k1;
    n.w = (uint32_t)p0;
    ctr = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return ctr;
}
// four N(0,1) draws for (seed, step, node, slot): two Box-Muller pairs
__device__ __forceinline__ void normal4(uint64_t seed, uint32_t step, uint32_t node, uint32_t slot, float out[4]) {
  const u4 r = philox4x32(u4{node, slot, step, 0x45474e4eu}, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float k = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)r.x + 0.5f) * k, u1 = ((float)r.y + 0.5f) * k;
  const float u2 = ((float)r.z + 0.5f) * k, u3 = ((float)r.w + 0.5f) * k;
  const float r0 = sqrtf(-2.0f * __logf(fmaxf(u0, 1e-37f))), r1 = sqrtf(-2.0f * __logf(fmaxf(u2, 1e-37f)));
  float s, c;
  __sincosf(6.283185307179586f * u1, &s, &c);
  out[0] = r0 * c; out[1] = r0 * s;
  __sincosf(6.283185307179586f * u3, &s, &c);
  out[2] = r1 * c; out[3] = r1 * s;
}

// block-wide sum of up to 3 values (fixed order -> deterministic)
__device__ __forceinline__ void block_sum3(float& a, float& b, float& c, float* red) {
  const int tid = threadIdx.x;
  red[tid] = a; red[kThreads + tid] = b; red[2 * kThreads + tid] = c;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (tid < w) {
      red[tid] += red[tid + w];
      red[kThreads + tid] += red[kThreads + tid + w];
      red[2 * kThreads + tid] += red[2 * kThreads + tid + w];
    }
    __syncthreads();
  }
  a = red[0]; b = red[kThreads]; c = red[2 * kThreads];
  __syncthreads();
}

// out[n][d] = (a[n][d] - b[n][d]) - mean over the graph (or over all nodes)   -- remove_mean
__global__ __launch_bounds__(kThreads) void center_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         int D, const int* __restrict__ graph_ptr, int N,
                                                         int per_graph, float* __restrict__ out) {
  __shared__ float red[3 * kThreads];
  const int lo = per_graph ? graph_ptr[blockIdx.x] : 0, hi = per_graph ? graph_ptr[blockIdx.x + 1] : N;
  const int cnt = hi - lo;
  for (int d0 = 0; d0 < D; d0 += 3) {
    float s[3] = {0.f, 0.f, 0.f};
    for (int n = lo + threadIdx.x; n < hi; n += kThreads)
      for (int j = 0; j < 3 && d0 + j < D; ++j) s[j] += a[(size_t)n * D + d0 + j] - (b ? b[(size_t)n * D + d0 + j] : 0.f);
    block_sum3(s[0], s[1], s[2], red);
    for (int n = lo + threadIdx.x; n < hi; n += kThreads)
      for (int j = 0; j < 3 && d0 + j < D; ++j)
        out[(size_t)n * D + d0 + j] = (a[(size_t)n * D + d0 + j] - (b ? b[(size_t)n * D + d0 + j] : 0.f)) - s[j] / (float)cnt;
  }
}

__global__ void slice_cols_kernel(const float* __restrict__ src, int ld, int A, int N, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N * A) dst[i] = src[(size_t)(i / A) * ld + i % A];
}

// z_out = z*c0 - eps*c1 + c2*noise'   (reverse_diffuse_one_step, diffusion_x_h.py:75-90)
__global__ __launch_bounds__(kThreads) void reverse_step_kernel(int D, int mode_pos, int per_graph, float c0, float c1,
                                                               float c2, const float* __restrict__ z, int ldz,
                                                               const float* __restrict__ eps,
                                                               const float* __restrict__ noise, float* __restrict__ zo,
                                                               int ldo, const int* __restrict__ graph_ptr, int N) {
  __shared__ float red[3 * kThreads];
  const int lo = per_graph ? graph_ptr[blockIdx.x] : 0, hi = per_graph ? graph_ptr[blockIdx.x + 1] : N;
  const int cnt = hi - lo;
  for (int d0 = 0; d0 < D; d0 += 3) {
    float m[3] = {0.f, 0.f, 0.f};
    if (mode_pos) {
      for (int n = lo + threadIdx.x; n < hi; n += kThreads)
        for (int j = 0; j < 3 && d0 + j < D; ++j) m[j] += noise[(size_t)n * D + d0 + j];
      block_sum3(m[0], m[1], m[2], red);
      for (int j = 0; j < 3; ++j) m[j] /= (float)cnt;
    }
    for (int n = lo + threadIdx.x; n < hi; n += kThreads)
      for (int j = 0; j < 3 && d0 + j < D; ++j) {
        const int d = d0 + j;
        const float mu = z[(size_t)n * ldz + d] * c0 - eps[(size_t)n * D + d] * c1;
        zo[(size_t)n * ldo + d] = mu + c2 * (noise[(size_t)n * D + d] - m[j]);
      }
  }
}

// ---- fused sampler kernels (one workgroup per graph) ----------------------------------------------
struct StepParams {
  int N, H, A, T;
  float scale;
  uint64_t seed;
  const int* graph_ptr;
  const float* table;   // [(T+1)][4]
  int* t_dev;           // current t (device-resident loop); null -> t_imm (caller-driven loop)
  unsigned* ticket;     // arrival counter of the step kernel's workgroups (device-resident loop)
  const int* t0_dev;    // t at the start of the run (indexes explicit noise)
  int t_imm;
  const float* noise_pos;  // [steps][N][3] or null
  const float* noise_h;    // [steps][N][A] or null
  const float* h_out;   // EGNN outputs
  const float* x_out;
  float* pos;           // state, updated in place
  float* h;             // state [N][H], columns [0,A) and H-1 updated in place
  int* bad;             // [B]
  int x_only;           // 1: the x-only loop of test.py:253-279 -- atom types stay fixed, only positions diffuse
};
constexpr int kMaxA = 8;

// one reverse step t -> t-1 (train_per_iretation.py:366-373): eps_x = remove_mean(x_L - pos),
// eps_h = h_L[:, :A]; pos <- mu(pos, eps_x) + std * remove_mean(noise); x <- mu(h[:, :A], eps_h) +
// std * noise; h[:, :A] <- scale * x; time column <- (t-1)/T.
__global__ __launch_bounds__(kThreads) void sampler_step_kernel(const StepParams p) {
  __shared__ float red[3 * kThreads];
  const int g = blockIdx.x, lo = p.graph_ptr[g], hi = p.graph_ptr[g + 1], cnt = hi - lo;
  const int t = p.t_dev ? *p.t_dev : p.t_imm;
  if (t < 1) return;
  const float c0 = p.table[4 * t], c1 = p.table[4 * t + 1], c2 = p.table[4 * t + 2];
  const float tnext = p.table[4 * (t - 1) + 3];
  const size_t nstep = p.t_dev ? (size_t)(*p.t0_dev - t) : 0;
  // pass 1: means of (x_out - pos) and of the position noise
  float e[3] = {0.f, 0.f, 0.f}, m[3] = {0.f, 0.f, 0.f};
  for (int n = lo + threadIdx.x; n < hi; n += kThreads) {
    float z[4];
    if (p.noise_pos) { for (int j = 0; j < 3; ++j) z[j] = p.noise_pos[(nstep * p.N + n) * 3 + j]; }
    else normal4(p.seed, (uint32_t)t, (uint32_t)n, 0u, z);
    for (int j = 0; j < 3; ++j) {
      e[j] += p.x_out[3 * n + j] - p.pos[3 * n + j];
      m[j] += z[j];
    }
  }
  block_sum3(e[0], e[1], e[2], red);
  block_sum3(m[0], m[1], m[2], red);
  int bad = 0;
  for (int n = lo + threadIdx.x; n < hi; n += kThreads) {
    float z[4];
    if (p.noise_pos) { for (int j = 0; j < 3; ++j) z[j] = p.noise_pos[(nstep * p.N + n) * 3 + j]; }
    else normal4(p.seed, (uint32_t)t, (uint32_t)n, 0u, z);
    for (int j = 0; j < 3; ++j) {
      const float pz = p.pos[3 * n + j];
      const float eps = (p.x_out[3 * n + j] - pz) - e[j] / (float)cnt;
      const float v = (pz * c0 - eps * c1) + c2 * (z[j] - m[j] / (float)cnt);
      p.pos[3 * n + j] = v;
      bad |= !isfinite(v);
    }
    for (int a0 = 0; a0 < (p.x_only ? 0 : p.A); a0 += 4) {
      float zh[4];
      if (!p.noise_h) normal4(p.seed, (uint32_t)t, (uint32_t)n, 1u + a0 / 4, zh);
      for (int j = 0; j < 4 && a0 + j < p.A; ++j) {
        const int a = a0 + j;
        const float nz = p.noise_h ? p.noise_h[(nstep * p.N + n) * p.A + a] : zh[j];
        const float hz = p.h[(size_t)n * p.H + a];
        const float v = (hz * c0 - p.h_out[(size_t)n * p.H + a] * c1) + c2 * nz;
        p.h[(size_t)n * p.H + a] = p.scale * v;
        bad |= !isfinite(v);
      }
    }
    p.h[(size_t)n * p.H + p.H - 1] = tnext;
  }
  if (bad) p.bad[g] = 1;
  // t <- t - 1 by the workgroup that finishes last: by then every workgroup has read t (no launch of its own)
  if (p.t_dev) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(p.ticket, 1u) == gridDim.x - 1) {
        *p.ticket = 0u;
        *p.t_dev = t - 1;
      }
    }
  }
}

__global__ void set_int_kernel(int* dst, int v, int* dst2, int v2) { *dst = v; if (dst2) *dst2 = v2; }

// x_T, h_T ~ N(0, I), positions mean-removed per graph (:301-305); h = [scale*x | cond | 1.0]
__global__ __launch_bounds__(kThreads) void sampler_init_kernel(const StepParams p, const float* __restrict__ cond,
                                                               const float* __restrict__ pos_init,
                                                               const float* __restrict__ x_init) {
  __shared__ float red[3 * kThreads];
  const int g = blockIdx.x, lo = p.graph_ptr[g], hi = p.graph_ptr[g + 1], cnt = hi - lo;
  const int C = p.H - p.A - 1;
  float m[3] = {0.f, 0.f, 0.f};
  for (int n = lo + threadIdx.x; n < hi; n += kThreads) {
    float z[4];
    if (pos_init) { for (int j = 0; j < 3; ++j) z[j] = pos_init[3 * n + j]; }
    else normal4(p.seed, (uint32_t)(p.T + 1), (uint32_t)n, 0u, z);
    for (int j = 0; j < 3; ++j) m[j] += z[j];
  }
  block_sum3(m[0], m[1], m[2], red);
  for (int n = lo + threadIdx.x; n < hi; n += kThreads) {
    float z[4];
    if (pos_init) { for (int j = 0; j < 3; ++j) z[j] = pos_init[3 * n + j]; }
    else normal4(p.seed, (uint32_t)(p.T + 1), (uint32_t)n, 0u, z);
    for (int j = 0; j < 3; ++j) p.pos[3 * n + j] = z[j] - m[j] / (float)cnt;
    for (int a0 = 0; a0 < p.A; a0 += 4) {
      float zh[4];
      if (!x_init) normal4(p.seed, (uint32_t)(p.T + 1), (uint32_t)n, 1u + a0 / 4, zh);
      for (int j = 0; j < 4 && a0 + j < p.A; ++j)
        p.h[(size_t)n * p.H + a0 + j] = p.scale * (x_init ? x_init[(size_t)n * p.A + a0 + j] : zh[j]);
    }
    for (int k = 0; k < C; ++k) p.h[(size_t)n * p.H + p.A + k] = cond[(size_t)n * C + k];
    p.h[(size_t)n * p.H + p.H - 1] = p.table[4 * p.T + 3];
  }
  if (threadIdx.x == 0) p.bad[g] = 0;
}

// final decode at t = 0 (train_per_iretation.py:412-428)
__global__ __launch_bounds__(kThreads) void sampler_final_kernel(const StepParams p, const float* __restrict__ npos,
                                                                const float* __restrict__ nh, float* __restrict__ pos_out,
                                                                float* __restrict__ hc_out, int* __restrict__ onehot) {
  __shared__ float red[3 * kThreads];
  const int g = blockIdx.x, lo = p.graph_ptr[g], hi = p.graph_ptr[g + 1], cnt = hi - lo;
  const float ia = p.table[0], sa = p.table[1];  // 1/alpha_0, sigma_0/alpha_0
  float e[3] = {0.f, 0.f, 0.f}, m[3] = {0.f, 0.f, 0.f};
  for (int n = lo + threadIdx.x; n < hi; n += kThreads) {
    float z[4];
    if (npos) { for (int j = 0; j < 3; ++j) z[j] = npos[3 * n + j]; }
    else normal4(p.seed, 0u, (uint32_t)n, 0u, z);
    for (int j = 0; j < 3; ++j) { e[j] += p.x_out[3 * n + j] - p.pos[3 * n + j]; m[j] += z[j]; }
  }
  block_sum3(e[0], e[1], e[2], red);
  block_sum3(m[0], m[1], m[2], red);
  int bad = 0;
  for (int n = lo + threadIdx.x; n < hi; n += kThreads) {
    float z[4];
    if (npos) { for (int j = 0; j < 3; ++j) z[j] = npos[3 * n + j]; }
    else normal4(p.seed, 0u, (uint32_t)n, 0u, z);
    for (int j = 0; j < 3; ++j) {
      const float pz = p.pos[3 * n + j];
      const float eps = (p.x_out[3 * n + j] - pz) - e[j] / (float)cnt;
      const float v = (pz * ia - sa * eps) + sa * (z[j] - m[j] / (float)cnt);
      pos_out[3 * n + j] = v;
      bad |= !isfinite(v);
    }
    float best = -INFINITY;
    int arg = 0;
    for (int a0 = 0; a0 < p.A; a0 += 4) {
      float zh[4];
      if (!nh) normal4(p.seed, 0u, (uint32_t)n, 1u + a0 / 4, zh);
      for (int j = 0; j < 4 && a0 + j < p.A; ++j) {
        const int a = a0 + j;
        const float nz = nh ? nh[(size_t)n * p.A + a] : zh[j];
        const float v = (p.h[(size_t)n * p.H + a] * ia - sa * p.h_out[(size_t)n * p.H + a]) + sa * nz;
        hc_out[(size_t)n * p.A + a] = v;
        bad |= !isfinite(v);
        if (v > best) { best = v; arg = a; }   // first maximum, as torch.argmax
      }
    }
    for (int a = 0; a < p.A; ++a) onehot[(size_t)n * p.A + a] = a == arg;
  }
  if (bad) p.bad[g] = 1;
}

__global__ void copy_types_kernel(const float* __restrict__ h, int H, int A, float inv_scale, int N, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N * A) out[i] = h[(size_t)(i / A) * H + i % A] * inv_scale;
}

static StepParams make_params(egnn_ctx* c) {
  Sampler& s = c->smp;
  StepParams p;
  p.N = c->N; p.H = c->H; p.A = s.A; p.T = s.T; p.scale = s.onehot_scale; p.seed = s.seed;
  p.graph_ptr = c->graph_ptr; p.table = s.d_table; p.t_dev = s.t_dev; p.t0_dev = s.t_dev + 1;
  p.ticket = reinterpret_cast<unsigned*>(s.t_dev + 2);
  p.noise_pos = nullptr; p.noise_h = nullptr; p.h_out = s.h_out; p.x_out = s.x_out; p.pos = s.pos; p.h = s.h;
  p.bad = s.bad; p.t_imm = 0; p.x_only = s.x_only;
  return p;
}

static int enqueue_step(egnn_ctx* c, hipStream_t st, int prec, int norm_scope, const float* npos, const float* nh) {
  Sampler& s = c->smp;
  int rc;
  const float *hc = s.h, *xc = s.pos;
  for (int l = 0; l < c->L; ++l) {
    float* ho = (l == c->L - 1) ? s.h_out : c->h_tmp[l & 1];
    float* xo = (l == c->L - 1) ? s.x_out : c->x_tmp[l & 1];
    if ((rc = launch_layer(c, st, l, prec, norm_scope, hc, xc, ho, xo, false, l + 1 < c->L))) return rc;
    hc = ho; xc = xo;
  }
  StepParams p = make_params(c);
  p.noise_pos = npos; p.noise_h = nh;
  hipLaunchKernelGGL(sampler_step_kernel, dim3(c->B), dim3(kThreads), 0, st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

static void graph_free(Sampler& s) {
  for (int i = 0; i < 2; ++i) {
    if (s.graph_exec[i]) { (void)hipGraphExecDestroy(s.graph_exec[i]); s.graph_exec[i] = nullptr; }
    if (s.graph[i]) { (void)hipGraphDestroy(s.graph[i]); s.graph[i] = nullptr; }
  }
  s.graph_prec = s.graph_norm = -1;
}

// hipGraph of `nsteps` consecutive reverse steps (the step index lives in device memory, so one graph serves any t)
static int graph_build(egnn_ctx* c, hipStream_t st, int prec, int norm_scope, int slot, int nsteps) {
  Sampler& s = c->smp;
  EGNN_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
  int rc = EGNN_OK;
  for (int i = 0; i < nsteps && !rc; ++i) rc = enqueue_step(c, st, prec, norm_scope, nullptr, nullptr);
  hipError_t e = hipStreamEndCapture(st, &s.graph[slot]);
  if (rc) return rc;
  if (e != hipSuccess) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return EGNN_EHIP; }
  EGNN_HIP(hipGraphInstantiate(&s.graph_exec[slot], s.graph[slot], nullptr, nullptr, 0));
  return EGNN_OK;
}

}  // namespace egnn

using namespace egnn;

extern "C" {

void sampler_free(egnn_ctx* c) {
  Sampler& s = c->smp;
  graph_free(s);
  void* ptrs[] = {s.pos, s.h, s.h_out, s.x_out, s.t_dev, s.bad, s.cond};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  s = Sampler();
}

int egnn_eps(void* stream, int N, int H, int A, const int32_t* graph_ptr, int B, const float* h_out,
             const float* x_out, const float* x_in, float* eps_x, float* eps_h) {
  if (N < 1 || !h_out || !x_out || !x_in || !eps_x || !eps_h || A < 1 || A > H || (graph_ptr && B < 1)) {
    set_error("bad egnn_eps arguments");
    return EGNN_EINVAL;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int pg = graph_ptr != nullptr;
  hipLaunchKernelGGL(center_kernel, dim3(pg ? B : 1), dim3(kThreads), 0, st, x_out, x_in, 3, graph_ptr, N, pg, eps_x);
  hipLaunchKernelGGL(slice_cols_kernel, dim3((N * A + 255) / 256), dim3(256), 0, st, h_out, H, A, N, eps_h);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egnn_remove_mean(void* stream, int N, int D, const int32_t* graph_ptr, int B, const float* in, float* out) {
  if (N < 1 || !in || !out || D < 1 || (graph_ptr && B < 1)) { set_error("bad egnn_remove_mean arguments"); return EGNN_EINVAL; }
  const int pg = graph_ptr != nullptr;
  hipLaunchKernelGGL(center_kernel, dim3(pg ? B : 1), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), in,
                     (const float*)nullptr, D, graph_ptr, N, pg, out);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int ddpm_reverse_step(void* stream, int N, int D, int mode_pos, const int32_t* graph_ptr, int B, float c0, float c1,
                      float c2, const float* z, int ldz, const float* eps, const float* noise, float* z_out, int ldo) {
  if (N < 1 || !z || !eps || !noise || !z_out || D < 1 || ldz < D || ldo < D || (graph_ptr && B < 1)) {
    set_error("bad ddpm_reverse_step arguments");
    return EGNN_EINVAL;
  }
  const int pg = graph_ptr != nullptr;
  hipLaunchKernelGGL(reverse_step_kernel, dim3(pg ? B : 1), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), D,
                     mode_pos, pg, c0, c1, c2, z, ldz, eps, noise, z_out, ldo, graph_ptr, N);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

// ---- the sampler's three fused kernels on CALLER-owned state (node-partitioned graphs: the EGNN forward between two
// steps is run stage by stage with collectives in between, so the loop is driven by the caller) ----
static int ext_params(StepParams& p, int N, int H, int A, int B, int T, const int32_t* graph_ptr, const float* table,
                      float scale, uint64_t seed, float* pos, float* h, int32_t* bad) {
  if (N < 1 || B < 1 || T < 1 || A < 1 || A > kMaxA || A + 1 > H || !graph_ptr || !table || !pos || !h || !bad) {
    set_error("bad ddpm_sampler_* arguments (N=%d B=%d T=%d A=%d H=%d)", N, B, T, A, H);
    return EGNN_EINVAL;
  }
  p.N = N; p.H = H; p.A = A; p.T = T; p.scale = scale; p.seed = seed; p.graph_ptr = graph_ptr; p.table = table;
  p.t_dev = nullptr; p.t0_dev = nullptr; p.ticket = nullptr; p.t_imm = 0; p.noise_pos = nullptr; p.noise_h = nullptr;
  p.h_out = nullptr; p.x_out = nullptr; p.pos = pos; p.h = h; p.bad = bad; p.x_only = 0;
  return EGNN_OK;
}

int ddpm_sampler_init(void* stream, int N, int H, int A, int B, int T, const int32_t* graph_ptr, const float* table,
                      float onehot_scale, uint64_t seed, const float* cond, const float* pos_init, const float* x_init,
                      float* pos, float* h, int32_t* bad) {
  StepParams p;
  int rc = ext_params(p, N, H, A, B, T, graph_ptr, table, onehot_scale, seed, pos, h, bad);
  if (rc) return rc;
  if (H - A - 1 > 0 && !cond) { set_error("conditioning block missing"); return EGNN_EINVAL; }
  hipLaunchKernelGGL(sampler_init_kernel, dim3(B), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), p, cond, pos_init,
                     x_init);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int ddpm_sampler_step(void* stream, int N, int H, int A, int B, int T, int t, const int32_t* graph_ptr, const float* table,
                      float onehot_scale, uint64_t seed, const float* h_out, const float* x_out, const float* noise_pos,
                      const float* noise_h, float* pos, float* h, int32_t* bad) {
  StepParams p;
  int rc = ext_params(p, N, H, A, B, T, graph_ptr, table, onehot_scale, seed, pos, h, bad);
  if (rc) return rc;
  if (t < 1 || t > T || !h_out || !x_out) { set_error("bad ddpm_sampler_step arguments (t=%d)", t); return EGNN_EINVAL; }
  p.t_imm = t; p.h_out = h_out; p.x_out = x_out; p.noise_pos = noise_pos; p.noise_h = noise_h;
  hipLaunchKernelGGL(sampler_step_kernel, dim3(B), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int ddpm_sampler_final(void* stream, int N, int H, int A, int B, int T, const int32_t* graph_ptr, const float* table,
                       float onehot_scale, uint64_t seed, const float* h_out, const float* x_out, const float* noise_pos,
                       const float* noise_h, float* pos, float* h, int32_t* bad, float* pos_out, float* hc_out,
                       int32_t* onehot_out) {
  StepParams p;
  int rc = ext_params(p, N, H, A, B, T, graph_ptr, table, onehot_scale, seed, pos, h, bad);
  if (rc) return rc;
  if (!h_out || !x_out || !pos_out || !hc_out || !onehot_out) { set_error("bad ddpm_sampler_final arguments"); return EGNN_EINVAL; }
  p.h_out = h_out; p.x_out = x_out;
  hipLaunchKernelGGL(sampler_final_kernel, dim3(B), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), p, noise_pos,
                     noise_h, pos_out, hc_out, onehot_out);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

// (schedule_table_build / schedule_table_from_alpha: host_logic.cpp)

// ---- sampler --------------------------------------------------------------------------------------
int egnn_sampler_prepare(egnn_ctx* c, int T, int A, float onehot_scale, const float* d_table, const float* d_cond,
                         uint64_t seed) {
  if (!c || c->N == 0 || c->L == 0) { set_error("model/graph not set"); return EGNN_ESTATE; }
  if (T < 1 || A < 1 || A > kMaxA || A + 1 > c->H || !d_table || (c->H - A - 1 > 0 && !d_cond)) {
    set_error("bad sampler arguments (T=%d A=%d H=%d)", T, A, c->H);
    return EGNN_EINVAL;
  }
  EGNN_HIP(hipSetDevice(c->device));
  EGNN_HIP(hipDeviceSynchronize());
  sampler_free(c);
  (void)fork_streams(c);   // small graphs: the message edge kernel runs beside the coordinate kernel
  Sampler& s = c->smp;
  const size_t N = c->N;
  EGNN_HIP(hipMalloc((void**)&s.pos, N * 3 * sizeof(float)));
  EGNN_HIP(hipMalloc((void**)&s.h, N * c->H * sizeof(float)));
  EGNN_HIP(hipMalloc((void**)&s.h_out, N * c->H * sizeof(float)));
  EGNN_HIP(hipMalloc((void**)&s.x_out, N * 3 * sizeof(float)));
  EGNN_HIP(hipMalloc((void**)&s.t_dev, 4 * sizeof(int)));   // {t, t at the start of the run, arrival ticket, -}
  EGNN_HIP(hipMemset(s.t_dev, 0, 4 * sizeof(int)));
  EGNN_HIP(hipMalloc((void**)&s.bad, (size_t)c->B * sizeof(int)));
  EGNN_HIP(hipMemset(s.bad, 0, (size_t)c->B * sizeof(int)));
  s.T = T; s.A = A; s.onehot_scale = onehot_scale; s.seed = seed; s.d_table = d_table; s.t = T;
  s.ready = true;
  if (c->H - A - 1 > 0) {
    const size_t bytes = N * (size_t)(c->H - A - 1) * sizeof(float);
    EGNN_HIP(hipMalloc((void**)&s.cond, bytes));
    EGNN_HIP(hipMemcpy(s.cond, d_cond, bytes, hipMemcpyDeviceToDevice));
  }
  return EGNN_OK;
}

int egnn_sampler_set_mode(egnn_ctx* c, int x_only) {
  if (!c || !c->smp.ready) { set_error("egnn_sampler_prepare first"); return EGNN_ESTATE; }
  if (x_only != 0 && x_only != 1) { set_error("sampler mode must be 0 (x and h) or 1 (x only)"); return EGNN_EINVAL; }
  if (c->smp.x_only != x_only) graph_free(c->smp);   // the captured step kernels carry the mode
  c->smp.x_only = x_only;
  return EGNN_OK;
}

int egnn_sampler_init(egnn_ctx* c, void* stream, const float* d_pos_init, const float* d_x_init) {
  if (!c || !c->smp.ready) { set_error("egnn_sampler_prepare first"); return EGNN_ESTATE; }
  if (c->smp.x_only && !d_x_init) { set_error("the x-only sampler needs the fixed atom types (x_init)"); return EGNN_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  Sampler& s = c->smp;
  s.t = s.T;
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, st, s.t_dev, s.T, s.t_dev + 1, s.T);
  StepParams p = make_params(c);
  hipLaunchKernelGGL(sampler_init_kernel, dim3(c->B), dim3(kThreads), 0, st, p, (const float*)s.cond, d_pos_init,
                     d_x_init);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egnn_sampler_run(egnn_ctx* c, void* stream, int prec, int norm_scope, int nsteps, int use_graph,
                     const float* d_noise_pos, const float* d_noise_h) {
  if (!c || !c->smp.ready) { set_error("egnn_sampler_prepare first"); return EGNN_ESTATE; }
  Sampler& s = c->smp;
  if (nsteps < 0 || nsteps > s.t) { set_error("nsteps %d exceeds remaining steps %d", nsteps, s.t); return EGNN_EINVAL; }
  if (nsteps == 0) return EGNN_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc;
  // t0 (start of this run) indexes the explicit noise arrays
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, st, s.t_dev + 1, s.t, (int*)nullptr, 0);
  if (use_graph && (d_noise_pos || d_noise_h)) { set_error("explicit noise is only supported without graph replay"); return EGNN_EINVAL; }
  if (use_graph) {
    if (st == nullptr) { set_error("hipGraph replay needs a non-default stream"); return EGNN_EINVAL; }
    if (c->prof) { set_error("disable profiling events before graph replay"); return EGNN_ESTATE; }
    if (s.graph_prec != prec || s.graph_norm != norm_scope) {
      graph_free(s);
      s.graph_prec = prec; s.graph_norm = norm_scope;
    }
    // one replay = kGraphSteps reverse steps (a replay costs the host ~10-16 us: amortised over 8 steps), single
    // steps for the remainder
    int left = nsteps;
    for (int slot = 1; slot >= 0; --slot) {
      const int per = slot ? kGraphSteps : 1;
      if (left < per) continue;
      if (!s.graph_exec[slot] && (rc = graph_build(c, st, prec, norm_scope, slot, per))) { graph_free(s); return rc; }
      for (; left >= per; left -= per) EGNN_HIP(hipGraphLaunch(s.graph_exec[slot], st));
    }
  } else {
    for (int i = 0; i < nsteps; ++i)
      if ((rc = enqueue_step(c, st, prec, norm_scope, d_noise_pos, d_noise_h))) return rc;
  }
  s.t -= nsteps;
  return EGNN_OK;
}

int egnn_sampler_final(egnn_ctx* c, void* stream, int prec, int norm_scope, const float* d_noise_pos,
                       const float* d_noise_h, float* d_pos_out, float* d_hc_out, int32_t* d_onehot_out) {
  if (!c || !c->smp.ready) { set_error("egnn_sampler_prepare first"); return EGNN_ESTATE; }
  Sampler& s = c->smp;
  if (s.t != 0) { set_error("final decode called at t=%d (must be 0)", s.t); return EGNN_ESTATE; }
  if (s.x_only) { set_error("the x-only loop (test.py:253-279) has no t = 0 decode: read egnn_sampler_state"); return EGNN_ESTATE; }
  if (!d_pos_out || !d_hc_out || !d_onehot_out) { set_error("null output"); return EGNN_EINVAL; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc;
  const float *hc = s.h, *xc = s.pos;
  for (int l = 0; l < c->L; ++l) {
    float* ho = (l == c->L - 1) ? s.h_out : c->h_tmp[l & 1];
    float* xo = (l == c->L - 1) ? s.x_out : c->x_tmp[l & 1];
    if ((rc = launch_layer(c, st, l, prec, norm_scope, hc, xc, ho, xo, false, l + 1 < c->L))) return rc;
    hc = ho; xc = xo;
  }
  StepParams p = make_params(c);
  hipLaunchKernelGGL(sampler_final_kernel, dim3(c->B), dim3(kThreads), 0, st, p, d_noise_pos, d_noise_h, d_pos_out,
                     d_hc_out, d_onehot_out);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egnn_sampler_state(egnn_ctx* c, void* stream, float* d_pos, float* d_x_types, int32_t* d_bad_flags, int* t_host) {
  if (!c || !c->smp.ready) { set_error("egnn_sampler_prepare first"); return EGNN_ESTATE; }
  Sampler& s = c->smp;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d_pos) EGNN_HIP(hipMemcpyAsync(d_pos, s.pos, (size_t)c->N * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (d_x_types)
    hipLaunchKernelGGL(copy_types_kernel, dim3((c->N * s.A + 255) / 256), dim3(256), 0, st, s.h, c->H, s.A,
                       1.0f / s.onehot_scale, c->N, d_x_types);
  if (d_bad_flags) EGNN_HIP(hipMemcpyAsync(d_bad_flags, s.bad, (size_t)c->B * sizeof(int), hipMemcpyDeviceToDevice, st));
  if (t_host) *t_host = s.t;
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // extern "C"
