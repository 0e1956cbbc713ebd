// Row-streaming product on the matrix cores: out[e][n] = sum_c A0[e][c] W0[c][n] (+ sum_c A1[e][c] W1[c][n]), n < 128,
// for E ~ 10^6 rows and c up to a few thousand -- the first-layer dgrad of the training backward,
//     dL/d[h_i | h_j | d2] = dL/da1x . W1x + dL/da1m . W1m        (EquivariantGraphNeuralNetwork.py:13, :19 under autograd),
// which round 2 ran as two library GEMMs (torch.mm + addmm_).  HBM-bound by construction (every operand row is read once:
// 4.2 GB per layer at the reference widths, 0.5 TFLOP), so the design is a streaming one:
//   * workgroup = 256 rows, wave w = rows 32 w .. 32 w + 31 x all 128 columns (4 accumulator tiles);
//   * 64 columns of c per step: buffer_load ... lds of [256 rows][128 B] (one 1-KiB wave instruction = 8 whole rows) into a
//     ring of 3 step buffers; the eight 16-byte pieces of a row are XOR-swizzled by (row >> 1) & 7 on the SOURCE side, which
//     makes the ds_read_b128 operand reads (32 rows, one piece each) conflict-free;
//   * W fragments packed [c / 64][128 / 32 x 4 k-steps][lane][8] (16 KiB per step, contiguous), staged by LDS-DMA too and
//     shared by the 8 waves: no register-destination load in the loop (hipcc drains every LDS-DMA in flight before the first
//     use of an ordinary load's result: cdna_hip_programming.md, "Three .s-level traps" (b));
//   * counted vmcnt + one raw s_barrier per step, as gemm_tn.hip.
#include "kernels.h"

namespace egnn {

struct GemmRowsParams {
  const void* A[2];     // bf16 [E][lda]
  int lda[2], K[2];     // K % 64 == 0 (K[1] = 0: one operand)
  const void* W[2];     // bf16 fragments [K / 64][4 k-steps][4 column blocks][64][8]: B[k = c][n]
  int E;
  void* out;            // bf16 (or fp32) [E][ldo], 128 columns written per column chunk (blockIdx.y): chunk j -> columns 128 j ..
  int ldo;              // (the W packs of chunk j follow those of chunk j - 1: K * 128 elements each)
  int out_f32;          // 1: fp32 output (node MLP products, where the result feeds a nonlinearity's derivative)
};

namespace {

typedef __attribute__((address_space(3))) void lds_void;
constexpr int kTR = 512, kRows = 256, kKS = 64, kNB = 4, kATileR = kRows * kKS * 2, kWTileR = 16 * 1024, kBufR = kATileR + kWTileR, kNBUF = 3;
constexpr size_t kSmemR = (size_t)kNBUF * kBufR;   // the store staging reuses the ring

__device__ __forceinline__ void dma16r(rsrc_t rs, char* dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, voff, 0, 0, 0);
}

__global__ __launch_bounds__(kTR, 2) void gemm_rows_kernel(const GemmRowsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int e0 = blockIdx.x * kRows, chunk = blockIdx.y;
  const int steps0 = p.K[0] / kKS, nsteps = steps0 + p.K[1] / kKS;
  const rsrc_t rs_a0 = make_rsrc(p.A[0], (unsigned)min((size_t)p.E * p.lda[0] * 2, (size_t)0xFFFFFFFFu));
  const rsrc_t rs_a1 = make_rsrc(p.A[1] ? p.A[1] : p.A[0], p.A[1] ? (unsigned)min((size_t)p.E * p.lda[1] * 2, (size_t)0xFFFFFFFFu) : 0u);
  const rsrc_t rs_w0 = make_rsrc(static_cast<const char*>(p.W[0]) + (size_t)chunk * 128 * p.K[0] * 2, (unsigned)((size_t)128 * p.K[0] * 2));
  const rsrc_t rs_w1 = make_rsrc(p.W[1] ? static_cast<const char*>(p.W[1]) + (size_t)chunk * 128 * p.K[1] * 2 : static_cast<const char*>(p.W[0]),
                                 (unsigned)((size_t)128 * (p.W[1] ? p.K[1] : p.K[0]) * 2));
  // staging: wave instruction i = wave + 8 k (k = 0..3) covers rows 8 i .. 8 i + 7 of the tile; lane -> (row, LDS piece)
  unsigned vrow[4], vpiece[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = 8 * (wave + 8 * k) + (lane >> 3);
    vrow[k] = (unsigned)(e0 + row);
    vpiece[k] = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
  }
  auto issue = [&](const int t) {   // 4 + 2 wave instructions per wave (always issued: the vmcnt arithmetic below counts them;
    const int tt = t < nsteps ? t : nsteps - 1;   // past the end: a harmless repeat of the last step into a free buffer)
    char* buf = smem + (size_t)(t % kNBUF) * kBufR;
    const bool second = tt >= steps0;
    const unsigned ld2 = (unsigned)(second ? p.lda[1] : p.lda[0]) * 2u, cs = (unsigned)(second ? tt - steps0 : tt);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma16r(second ? rs_a1 : rs_a0, buf + (wave + 8 * k) * 1024, vrow[k] * ld2 + cs * (kKS * 2) + vpiece[k]);
#pragma unroll
    for (int k = 0; k < 2; ++k) dma16r(second ? rs_w1 : rs_w0, buf + kATileR + (wave + 8 * k) * 1024, cs * kWTileR + (wave + 8 * k) * 1024 + lane * 16u);
  };
  // operand reads: lane (row r of the wave's 32, half hh) takes piece 2 s + hh of its row for k-step s
  const int arow = 32 * wave + r;
  unsigned aoff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) aoff[s] = (unsigned)(arow * 128 + (((2 * s + hh) ^ ((arow >> 1) & 7)) << 4));

  f32x16 acc[kNB];
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

#define VM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  issue(0);
  issue(1);
  for (int s = 0; s < nsteps; ++s) {
    VM_WAIT(6);                                // all but the youngest step's 6 DMAs: this wave's pieces of step s have landed
    __builtin_amdgcn_s_barrier();              // ... every wave's; and every wave has finished step s - 1
    issue(s + 2);                              // refill of buffer (s - 1) % 3 (WAR: behind the barrier)
    const char* bufp = smem + (size_t)(s % kNBUF) * kBufR;
    bf16x8 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const bf16x8*>(bufp + aoff[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bf16x8 bw[kNB];
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb) bw[nb] = *reinterpret_cast<const bf16x8*>(bufp + kATileR + (k * kNB + nb) * 1024 + lane * 16);
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], bw[nb], acc[nb], 0, 0, 0);
    }
  }
  VM_WAIT(0);
  __builtin_amdgcn_s_barrier();                // the ring is free: reused as store staging
#undef VM_WAIT
  if (p.out_f32) {   // accumulator layout straight to memory: 128 contiguous bytes per half-wave and row
    float* out = static_cast<float*>(p.out) + (size_t)(e0 + 32 * wave) * p.ldo + 128 * chunk;
    const int nrows = p.E - (e0 + 32 * wave);
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = acc_row(i, lane);
        if (row < nrows) out[(size_t)row * p.ldo + 32 * nb + r] = acc[nb][i];
      }
    return;
  }
  // ---- store: 32 rows x 128 columns per wave as row-major bf16 (two 64-column halves through the per-wave staging tile) ----
  __bf16* stg = reinterpret_cast<__bf16*>(smem) + (size_t)wave * 32 * 72;
  __bf16* out = static_cast<__bf16*>(p.out) + (size_t)(e0 + 32 * wave) * p.ldo + 128 * chunk;
  const int nrows = p.E - (e0 + 32 * wave);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x16 blk[2];
    blk[0] = acc[2 * half];
    blk[1] = acc[2 * half + 1];
    store_block_bf16(blk, 2, stg, out + 64 * half, (size_t)p.ldo, nrows, lane);
  }
}

}  // namespace

int launch_gemm_rows(const GemmRowsParams& p, int nchunks, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_rows_kernel, dim3((p.E + kRows - 1) / kRows, nchunks), dim3(kTR), kSmemR, st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

// W [K][ldw] row-major fp32 (first `ncols` columns used, the rest of the last 128-column chunk zero) -> bf16 B fragments, per
// chunk (blockIdx.y) [K / 64 steps][4 k-steps][4 column blocks][64 lanes][8]: lane l holds
// B[k = 64 step + 16 ks + 8 (l >> 5) + j][n = 128 chunk + 32 nb + (l & 31)]
__global__ void pack_rows_weights_kernel(const float* __restrict__ W, int K, int ldw, int ncols, __bf16* __restrict__ out) {
  const size_t total = (size_t)K * 128;
  W += 128 * blockIdx.y;
  ncols -= 128 * blockIdx.y;
  out += total * blockIdx.y;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63;
    const size_t f = i >> 9;                 // fragment index = (step * 4 + ks) * 4 + nb
    const int nb = f & 3, ks = (f >> 2) & 3, step = (int)(f >> 4);
    const int n = 32 * nb + (lane & 31), k = 64 * step + 16 * ks + 8 * (lane >> 5) + j;
    out[i] = (__bf16)(n < ncols ? W[(size_t)k * ldw + n] : 0.f);
  }
}
int launch_pack_rows_weights(const float* W, int K, int ldw, int ncols, void* out, hipStream_t st) {
  hipLaunchKernelGGL(pack_rows_weights_kernel, dim3(128, (ncols + 127) / 128), dim3(256), 0, st, W, K, ldw, ncols, static_cast<__bf16*>(out));
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace egnn

using namespace egnn;

extern "C" {

int egnn_gemm_rows_pack(void* stream, int K, int ncols, const float* d_W, int ldw, void* d_frags_out) {
  if (K < 64 || K % 64 != 0 || ncols < 1 || ncols > 128 * 65535 || ldw < ncols || !d_W || !d_frags_out) {
    set_error("egnn_gemm_rows_pack: need K %% 64 == 0, ncols >= 1");
    return EGNN_EINVAL;
  }
  return launch_pack_rows_weights(d_W, K, ldw, ncols, d_frags_out, reinterpret_cast<hipStream_t>(stream));
}

int egnn_gemm_rows_bf16(void* stream, int E, const void* d_A0, int lda0, int K0, const void* d_W0, const void* d_A1, int lda1, int K1,
                        const void* d_W1, void* d_out, int ldo, int out_f32, int n_chunks) {
  {
    const int rc = gemm_rows_args_check(E, d_A0, lda0, K0, d_W0, d_A1, lda1, K1, d_W1, d_out, ldo);   // host_logic.cpp
    if (rc) return rc;
    if (n_chunks < 1 || n_chunks > 65535 || ldo < 128 * n_chunks) { set_error("egnn_gemm_rows_bf16: %d column chunks need ldo >= %d", n_chunks, 128 * n_chunks); return EGNN_EINVAL; }
  }
  GemmRowsParams p;
  p.A[0] = d_A0; p.lda[0] = lda0; p.K[0] = K0; p.W[0] = d_W0;
  p.A[1] = d_A1; p.lda[1] = d_A1 ? lda1 : 0; p.K[1] = d_A1 ? K1 : 0; p.W[1] = d_A1 ? d_W1 : nullptr;
  p.E = E; p.out = d_out; p.ldo = ldo; p.out_f32 = out_f32 ? 1 : 0;
  return launch_gemm_rows(p, n_chunks, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
