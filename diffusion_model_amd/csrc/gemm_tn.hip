// Reductions over ALL edges (or nodes) on the matrix cores: C[M][N] = scale * sum_e A[e][m] * B[e][n], A and B row-major
// bf16 with the long dimension e as rows -- the weight gradients of the training backward (loss.backward() of
// parts/train_per_iretation.py:172 over EquivariantGraphNeuralNetwork.py:13-30):
//     dL/dW2 = dL/da2^T . s1        (mlp_x.2 / mlp_m.2: [E, 1024 | 256]^T x [E, 1024], E ~ 10^6)
//     dL/dW1 = dL/da1^T . [h_i | h_j | d2 | 1]   (first layers, 80 padded to 128 columns),     node MLP likewise over nodes.
// Round 2 ran these on the BLAS library (torch.mm / bmm); this is the hand-written replacement.
//
// Both operands are "transposed" for the matrix instruction (the reduction index is the ROW index in memory), so the
// fragments come from ds_read_b64_tr_b16: the tiles are staged exactly as they lie in memory (rows of 512 / 256 bytes,
// LDS-DMA, no register hop) and the hardware transposes 4 x 16 blocks on the way to the registers.
//   * workgroup tile 256 (m) x BN (n) of C, BN = 256 (8 waves as 2 x 4, 128 x 64 each) or 128 (4 x 2, 64 x 64 each);
//   * the reduction is cut into slices (split-K): workgroup (slice, tile) writes its fp32 partial tile to a slab, a second
//     launch adds the slabs in slice order (deterministic, no atomics).  The tiles of one slice are neighbours in the
//     XCD-aware order, so the A and B rows of a slice are fetched once per XCD and shared through its L2;
//   * 32 rows of e per step; buffer_load ... lds (1 KiB per wave instruction) into a ring of 5 (6) step buffers, the
//     whole 160 KiB of LDS; the 16-byte pieces of a row are XOR-swizzled by the row's low two bits ON THE SOURCE SIDE (the DMA
//     writes LDS linearly), which makes every 32-lane half of a transposed read (4 rows x 64 bytes) conflict-free;
//   * counted vmcnt + one raw s_barrier per step (cdna_hip_programming.md, "Pipelining across barriers"): the DMA of step
//     s + NBUF - 1 is issued behind the barrier of step s and must have landed NBUF - 2 steps later; fragments of the next half step
//     are read while the current half step's MFMAs run (two fragment sets).
#include "kernels.h"

namespace egnn {

struct GemmTnParams {
  const void* A;   // bf16 [E][lda]
  const void* B;   // bf16 [E][ldb]
  int lda, ldb, E, M, N;
  int steps_per_slice, nslices, tiles_n;
  float* slabs;    // [nslices][M][N]
};

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;

constexpr int kTG = 512, kBM = 256, kBK = 32;
static_assert(kBM == kGemmBM && kBK == kGemmBK, "host_logic.cpp plans the slices for this tile");

// buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes from per-lane offsets `voff` of the buffer to LDS at `dst` + 16 * lane
// (dst wave-uniform).  A plain (non-template) device function: inside the kernel TEMPLATE the builtin makes the host pass
// drop the instantiation's stub without a diagnostic (hipcc 7.2).
__device__ __forceinline__ void dma16(rsrc_t rs, char* dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, voff, 0, 0, 0);
}

template <int BN>
__global__ __launch_bounds__(kTG, 2) void gemm_tn_kernel(const GemmTnParams p) {
  constexpr int RM = BN == 256 ? 4 : 2, RN = 2;          // 32 x 32 accumulator tiles per wave
  constexpr int WN = BN == 256 ? 4 : 2;                  // waves along n (waves along m = 8 / WN)
  constexpr int RBA = kBM * 2, RBB = BN * 2;             // bytes per LDS row
  constexpr int kATile = kBK * RBA, kBTile = kBK * RBB, kBuf = kATile + kBTile;
  constexpr int NBUF = (160 * 1024) / kBuf;              // 5 (BN = 256) / 6 (BN = 128)
  constexpr int DMA_A = 2, DMA_B = BN == 256 ? 2 : 1;    // wave instructions per wave and step
  constexpr int DMA = DMA_A + DMA_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = (p.M / kBM) * p.tiles_n;
  const int j = xcd_tile(blockIdx.x, gridDim.x);
  const int slice = j / ntiles, tile = j - slice * ntiles;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m_tile0 = tm * kBM, n_tile0 = tn * BN;
  const int step0 = slice * p.steps_per_slice;
  const int total_steps = (p.E + kBK - 1) / kBK;
  const int nsteps = min(p.steps_per_slice, total_steps - step0);
  const int wm = wave / WN, wn = wave - wm * WN;

  // ---- staging: per-lane source offsets (bytes) inside a step's rows; rows past E read as zero (buffer range check) ----
  const rsrc_t rs_a = make_rsrc(p.A, (unsigned)min((size_t)p.E * p.lda * 2, (size_t)0xFFFFFFFFu));
  const rsrc_t rs_b = make_rsrc(p.B, (unsigned)min((size_t)p.E * p.ldb * 2, (size_t)0xFFFFFFFFu));
  // A: wave instruction a = wave + 8 i (i = 0, 1) covers rows 2 a, 2 a + 1 of the step: lane -> (row, LDS piece)
  unsigned va[DMA_A], vb[DMA_B];
#pragma unroll
  for (int i = 0; i < DMA_A; ++i) {
    const int row = 2 * (wave + 8 * i) + (lane >> 5), piece = (lane & 31) ^ ((row & 3) << 2);
    va[i] = (unsigned)(((size_t)row * p.lda + m_tile0) * 2 + piece * 16);
  }
#pragma unroll
  for (int i = 0; i < DMA_B; ++i) {
    int row, piece;
    if constexpr (BN == 256) { row = 2 * (wave + 8 * i) + (lane >> 5); piece = (lane & 31) ^ ((row & 3) << 2); }
    else { row = 4 * wave + (lane >> 4); piece = (lane & 15) ^ ((row & 3) << 2); }
    vb[i] = (unsigned)(((size_t)row * p.ldb + n_tile0) * 2 + piece * 16);
  }
  auto issue = [&](const int t) {   // step t of this slice -> ring buffer t % NBUF
    char* buf = smem + (size_t)(t % NBUF) * kBuf;
    const unsigned ra = (unsigned)((size_t)(step0 + t) * kBK * p.lda * 2), rb_ = (unsigned)((size_t)(step0 + t) * kBK * p.ldb * 2);
#pragma unroll
    for (int i = 0; i < DMA_A; ++i)
      dma16(rs_a, buf + (wave + 8 * i) * 1024, va[i] + ra);
#pragma unroll
    for (int i = 0; i < DMA_B; ++i)
      dma16(rs_b, buf + kATile + (BN == 256 ? (wave + 8 * i) : wave) * 1024, vb[i] + rb_);
  };

  // ---- fragment addresses: ds_read_b64_tr_b16, lane 4 q + pp of a 16-lane group supplies row q, columns 4 pp .. 4 pp + 3 ----
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned aaddr[RM], baddr[RN];
#pragma unroll
  for (int rb = 0; rb < RM; ++rb) {
    const int piece = ((wm * 32 * RM + 32 * rb) >> 3) + 2 * (g & 1) + (pp >> 1);
    aaddr[rb] = (unsigned)((8 * (g >> 1) + q) * RBA + ((piece ^ (q << 2)) << 4) + (pp & 1) * 8);
  }
#pragma unroll
  for (int cb = 0; cb < RN; ++cb) {
    const int piece = ((wn * 64 + 32 * cb) >> 3) + 2 * (g & 1) + (pp >> 1);
    baddr[cb] = (unsigned)(kATile + (8 * (g >> 1) + q) * RBB + ((piece ^ (q << 2)) << 4) + (pp & 1) * 8);
  }

  f32x16 acc[RM][RN];
#pragma unroll
  for (int rb = 0; rb < RM; ++rb)
#pragma unroll
    for (int cb = 0; cb < RN; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;

  // fragment sets: [0] = first half step (rows 0..15 of the step), [1] = second half (rows 16..31)
  u32x2 fa[2][RM][2], fb[2][RN][2];
#define TR_RD(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  auto read_set = [&](const int hs, const unsigned bufoff) {   // hs must be a literal 0 / 1 at every call site
#pragma unroll
    for (int rb = 0; rb < RM; ++rb) {
      const unsigned ad = lds0 + bufoff + aaddr[rb];
      if (hs == 0) { TR_RD(fa[0][rb][0], ad, 0); TR_RD(fa[0][rb][1], ad, 4 * RBA); }
      else { TR_RD(fa[1][rb][0], ad, 16 * RBA); TR_RD(fa[1][rb][1], ad, 20 * RBA); }
    }
#pragma unroll
    for (int cb = 0; cb < RN; ++cb) {
      const unsigned ad = lds0 + bufoff + baddr[cb];
      if (hs == 0) { TR_RD(fb[0][cb][0], ad, 0); TR_RD(fb[0][cb][1], ad, 4 * RBB); }
      else { TR_RD(fb[1][cb][0], ad, 16 * RBB); TR_RD(fb[1][cb][1], ad, 20 * RBB); }
    }
  };
  constexpr int kReads = 2 * (RM + RN);   // LDS reads of one fragment set
  auto mfma_set = [&](const int hs) {
#pragma unroll
    for (int rb = 0; rb < RM; ++rb) asm volatile("" : "+v"(fa[hs][rb][0]), "+v"(fa[hs][rb][1]));   // uses stay below the wait
#pragma unroll
    for (int cb = 0; cb < RN; ++cb) asm volatile("" : "+v"(fb[hs][cb][0]), "+v"(fb[hs][cb][1]));
    bf16x8 bfr[RN];
#pragma unroll
    for (int cb = 0; cb < RN; ++cb)
      bfr[cb] = __builtin_bit_cast(bf16x8, u32x4v{fb[hs][cb][0].x, fb[hs][cb][0].y, fb[hs][cb][1].x, fb[hs][cb][1].y});
#pragma unroll
    for (int rb = 0; rb < RM; ++rb) {
      const bf16x8 afr = __builtin_bit_cast(bf16x8, u32x4v{fa[hs][rb][0].x, fa[hs][rb][0].y, fa[hs][rb][1].x, fa[hs][rb][1].y});
#pragma unroll
      for (int cb = 0; cb < RN; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr[cb], acc[rb][cb], 0, 0, 0);
    }
  };
#define VM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define LGKM_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")

  // ---- prologue: steps 0 .. NBUF - 2 in flight, step 0 landed, first fragment set read ----
#pragma unroll
  for (int t = 0; t < NBUF - 1; ++t) issue(t);
  VM_WAIT(DMA * (NBUF - 2));
  __builtin_amdgcn_s_barrier();
  read_set(0, 0u);

  for (int s = 0; s < nsteps; ++s) {
    VM_WAIT(DMA * (NBUF - 3));            // this wave's pieces of step s + 1 have landed (steps s + 2 ... may still fly)
    __builtin_amdgcn_s_barrier();         // ... and every other wave's; every wave has also finished step s - 1, i.e. its
    issue(s + NBUF - 1);                  // reads of buffer (s - 1) % NBUF, which is refilled only now (WAR)
    const unsigned cur = (unsigned)((s % NBUF) * kBuf), nxt = (unsigned)(((s + 1) % NBUF) * kBuf);
    read_set(1, cur);                     // second half of step s, under the first half's MFMAs
    LGKM_WAIT(kReads);                    // LDS returns in order: all but the kReads younger reads have landed
    mfma_set(0);
    read_set(0, nxt);                     // first half of step s + 1
    LGKM_WAIT(kReads);
    mfma_set(1);
  }
  VM_WAIT(0);   // drain the DMAs that ran ahead of the slice's end (LDS must not be written after the workgroup retires)
  LGKM_WAIT(0);
#undef TR_RD
#undef VM_WAIT
#undef LGKM_WAIT

  // ---- partial tile to this slice's slab ----
  float* slab = p.slabs + (size_t)slice * p.M * p.N;
  const int r = lane & 31;
#pragma unroll
  for (int rb = 0; rb < RM; ++rb)
#pragma unroll
    for (int cb = 0; cb < RN; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m_tile0 + wm * 32 * RM + 32 * rb + acc_row(i, lane), n = n_tile0 + wn * 64 + 32 * cb + r;
        slab[(size_t)m * p.N + n] = acc[rb][cb][i];
      }
}

// out[i] = (accumulate ? out[i] : 0) + scale * sum over slices (in slice order) of slabs[s][i], for the first `rows` x `cols`
// entries of each [M][N] slab (padding rows / columns of the operands are dropped)
__global__ void gemm_tn_reduce_kernel(const float* __restrict__ slabs, int S, int M, int N, int rows, int cols, float scale,
                                      float* __restrict__ out, int ldo, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int m = i / cols, n = i - m * cols;
  float v = 0.f;
  for (int s = 0; s < S; ++s) v += slabs[((size_t)s * M + m) * N + n];
  float* o = out + (size_t)m * ldo + n;
  *o = (accumulate ? *o : 0.f) + scale * v;
}

}  // namespace

int init_gemm_tn_attributes() {
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return EGNN_OK;
}

// launches (inside the namespace: the kernels live in its anonymous part)
int launch_gemm_tn(const GemmTnParams& p, int BN, int rows, int cols, float scale, float* C, int ldc, int accumulate, hipStream_t st) {
  const int grid = p.nslices * (p.M / kBM) * p.tiles_n;
  if (BN == 256) hipLaunchKernelGGL(gemm_tn_kernel<256>, dim3(grid), dim3(kTG), 160 * 1024, st, p);
  else hipLaunchKernelGGL(gemm_tn_kernel<128>, dim3(grid), dim3(kTG), 6 * (kBK * kBM * 2 + kBK * 128 * 2), st, p);
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, st, p.slabs, p.nslices, p.M, p.N, rows, cols,
                     scale, C, ldc, accumulate);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace egnn

using namespace egnn;

extern "C" {

int egnn_gemm_tn_bf16(void* stream, int E, int M, int N, const void* d_A, int lda, const void* d_B, int ldb, float scale,
                      float* d_C, int ldc, int rows, int cols, int accumulate, void* d_workspace, size_t workspace_bytes) {
  {   // shape / pointer / 4 GiB-offset / workspace checks and the split-K plan: host_logic.cpp
    const int rc = gemm_tn_args_check(E, M, N, d_A, lda, d_B, ldb, d_C, ldc, rows, cols, d_workspace, workspace_bytes);
    if (rc) return rc;
  }
  static bool attr_done = false;
  if (!attr_done) { int rc = init_gemm_tn_attributes(); if (rc) return rc; attr_done = true; }
  GemmTnParams p;
  int BN;
  plan_gemm_tn(E, M, N, BN, p.tiles_n, p.nslices, p.steps_per_slice);
  p.A = d_A; p.B = d_B; p.lda = lda; p.ldb = ldb; p.E = E; p.M = M; p.N = N;
  p.slabs = static_cast<float*>(d_workspace);
  return launch_gemm_tn(p, BN, rows, cols, scale, d_C, ldc, accumulate, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
