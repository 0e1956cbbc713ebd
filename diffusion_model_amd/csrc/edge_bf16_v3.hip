// Fused per-edge kernels, bf16 MFMA, 128-edge tiles with the output columns split over workgroups (gfx950).
//
// Why: at a 64-edge tile every CU must stream 40 KiB of weight fragments per 640 MFMA-cycles = 64 B/clk,
// the per-CU L2 fill rate, and the tile cannot grow because the fp32 accumulators of all 1280 output
// columns (mlp_x.2: 1024, mlp_m.2: 256) already fill the register file.  Here one 128-edge tile is
// processed by three workgroups, each holding only a slice of the columns:
//
//     X half 0 / X half 1 : mlp_x (EquivariantGraphNeuralNetwork.py:19-25, :62-65) columns [0,512) / [512,1024)
//     M                   : mlp_m + attention gate (:13-18, :31-34, :55-61), all 256 columns
//
// Every weight fragment now feeds 4 row blocks (128 edges) instead of 2, halving the weight bytes per
// MFMA; each workgroup rebuilds the SiLU(P[dst] + Q[src] + wd*d2) activations it needs (mlp_x's are
// built twice -- the price of not exchanging them between workgroups).  s_ij is linear in the column
// halves, so the two X workgroups write separate coordinate sums that node_post adds: no inter-workgroup
// communication, bitwise deterministic.
//
// Workgroup = 8 wave64 (2 per SIMD), wave w owns 32*CB columns; per 16-deep k-step: 4 A fragments from
// the double-buffered LDS activation chunk, CB B fragments straight from the packed weights, 4*CB MFMAs.
#include <type_traits>

#include "diag.h"
#include "edge_tile.h"

namespace egnn {

namespace {

constexpr int kT3 = 512;
constexpr int kR3 = 128, kRB3 = 4, kRPAD3 = kR3 + 1;
constexpr int kKC3 = 64;
constexpr size_t kA1_3 = (size_t)8 * kRPAD3 * 16;  // one activation chunk [8 k-groups][129][8 bf16]
using namespace tile128;
constexpr size_t kOffA1 = kOffLoop;      // 2 activation chunks, then wd[KP] (the per-tile arrays: edge_tile.h)
__host__ __device__ inline size_t v3_smem_bytes(int KP, int MP, bool is_m) {
  (void)MP; (void)is_m;
  return kOffA1 + 2 * kA1_3 + (size_t)KP * 4;
}

// BWD = true: the training backward's recompute pass over a chunk of edges (egcl_backward_edge_recompute).  Same
// prologue and K loop; the activation chunks are also written to HBM (s1_out: the wgrad of mlp_x.2 needs them as a GEMM
// operand over ALL edges), and the epilogue turns the accumulators into dL/d(a2) instead of segment sums.
// SAVE = true: the training FORWARD (egcl_forward_save): the forward kernel as it is, which also leaves the activation
// chunks (s1_out) and the scaled second-layer pre-activations -log2(e) * (a2 + b2) (g_a2_out) in HBM, so that the backward
// needs no recompute pass (egcl_backward_heads_saved turns the pre-activations into dL/da2 in place).
template <int CB, bool IS_M, bool BWD = false, bool SAVE = false>
__global__ __launch_bounds__(kT3, 2) void edge_kernel_bf16_v3(const EdgeParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lds L(smem);
  int* const s_dst = L.dst; int* const s_src = L.src;
  float* const s_d2 = L.d2; float* const s_diff = L.diff; float* const s_val = L.val; float* const s_part = L.part;
  float* const s_gseg = L.gseg;
  char* s_a1 = smem + kOffA1;
  float* s_wd = reinterpret_cast<float*>(s_a1 + 2 * kA1_3);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int KP = IS_M ? p.WmP : p.WxP;
  const int nsplit = IS_M ? 1 : p.WxP / (256 * CB);
  const int j = xcd_tile(blockIdx.x, gridDim.x);
  const int tile = j / nsplit, half = j - tile * nsplit;
  const int e0 = tile * kR3;
  const int nvalid = min(kR3, p.E - e0);

  DIAG_STAMP_SETUP(p.stamps + ((size_t)(IS_M ? 1 : 0) * 8 + wave) * 32 * 4);
  DIAG_STAMP(30, 0);   // kernel entry
  const int S = prologue(p, L, e0, nvalid, IS_M ? p.wdm : p.wdx, KP, s_wd, tid, lane, wave);
  if constexpr (BWD && !IS_M) {   // dL/ds_e = dL/d(sum_x[i]) . (x_i - x_j)   (:64, xm = (x_i - x_j) * s); read behind the K loop's barriers
    if (tid < kR3) {
      const int d = s_dst[tid];
      s_gseg[tid] = tid < nvalid ? (p.g_sum_x[3 * d] * s_diff[tid] + p.g_sum_x[3 * d + 1] * s_diff[kR3 + tid] +
                                    p.g_sum_x[3 * d + 2] * s_diff[2 * kR3 + tid]) : 0.f;
    }
  }

  DIAG_STAMP(30, 1);   // tile structure ready

  // ---- K-loop ----
  const int NC = KP / kKC3, KS = KP / 16;
  const int brow = tid >> 3, kg = tid & 7;   // this thread builds rows brow and brow + 64
  const rsrc_t rs_tab = make_rsrc(p.table, diag::drop_table_loads(p.dbg) ? 0u : (unsigned)((size_t)p.N * p.TC * 2));
  const rsrc_t rs_w = make_rsrc(IS_M ? p.w2m : p.w2x, diag::drop_weight_loads(p.dbg) ? 0u : (unsigned)((size_t)(IS_M ? p.MP : p.WxP) * KP * 2));
  const unsigned vdst0 = (unsigned)s_dst[brow] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const unsigned vsrc0 = (unsigned)s_src[brow] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const unsigned vdst1 = (unsigned)s_dst[brow + 64] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const unsigned vsrc1 = (unsigned)s_src[brow + 64] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const float d2r0 = s_d2[brow], d2r1 = s_d2[brow + 64];
  const unsigned offP = (IS_M ? 2u * p.WxP : 0u) * 2u, offQ = (IS_M ? 2u * p.WxP + p.WmP : (unsigned)p.WxP) * 2u;   // fp16 table
  char* slot0 = s_a1 + ((size_t)kg * kRPAD3 + brow) * 16;
  char* slot1 = slot0 + 64 * 16;
  const unsigned lane16 = lane * 16u;
  // 32-bit LDS byte address of this lane's A-fragment slot in activation buffer 0
  const unsigned lds_a1_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_a1 + ((size_t)hh * kRPAD3 + r) * 16);
  const int colblk0 = half * 8 * CB + wave * CB;   // first 32-column block of this wave
  const unsigned w0 = (unsigned)colblk0 * KS * 1024u;

  f32x16 acc[kRB3][CB];
#pragma unroll
  for (int rb = 0; rb < kRB3; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;

  // backward: the activation chunk also goes to HBM, 16 bytes per thread, 128 contiguous bytes per row and chunk
  // (the column-split coordinate workgroups build the same activations: only share 0 stores them)
  auto s1_store = [&](const bf16x8 o0, const bf16x8 o1, const int c) {
    if (half != 0) return;
    __bf16* base = static_cast<__bf16*>(p.s1_out) + (size_t)e0 * KP + c * kKC3 + kg * 8;
    if (brow < nvalid) *reinterpret_cast<bf16x8*>(base + (size_t)brow * KP) = o0;
    if (brow + 64 < nvalid) *reinterpret_cast<bf16x8*>(base + (size_t)(brow + 64) * KP) = o1;
  };
  {  // chunk 0
    UnitH u;
    unith_load(u, rs_tab, vdst0, vsrc0, offP, offQ);
    const bf16x8 o0 = unith_finish(u, s_wd + kg * 8, d2r0, slot0);
    unith_load(u, rs_tab, vdst1, vsrc1, offP, offQ);
    const bf16x8 o1 = unith_finish(u, s_wd + kg * 8, d2r1, slot1);
    if constexpr (BWD || SAVE) s1_store(o0, o1, 0);
  }
  bf16x8 bq[4][CB];   // weight fragments of the 4 k-steps of the current chunk
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) bq[s][cb] = ldbuf_bf16x8(rs_w, lane16, w0 + ((unsigned)cb * KS + s) * 1024u);
  __syncthreads();

  // matrix phase of chunk c: 4 k-steps x (4 row blocks x CB column blocks)
  // Operand pipeline of the matrix phase: the weight fragments of k-step s of chunk c+1 are requested right
  // after the MFMAs of k-step s of chunk c were issued (a whole chunk = 4 k-steps of distance, enough to cover
  // an L2 round trip under load), the LDS A fragments one k-step ahead.
  auto mphase = [&](const int c, const bool last) {
    // A fragments by inline-asm ds_read_b128 so that hipcc cannot sink them to their use: a[rb] is refilled in
    // place for k-step s+1 right after the MFMAs of (s, rb) were issued and flies under the next 3 row blocks'
    // MFMAs.  LDS operations of a wave return in order, so lgkmcnt(3) before a use means "all but the 3
    // younger reads have landed" (a scalar load in flight only makes the wait conservative).
    const unsigned abase = lds_a1_base + (unsigned)(c & 1) * (unsigned)kA1_3;
    bf16x8 a[kRB3];
#define LDS_RD(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(abase), "n"(off))
#define LDS_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
    LDS_RD(a[0], 0); LDS_RD(a[1], 512); LDS_RD(a[2], 1024); LDS_RD(a[3], 1536);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int rb = 0; rb < kRB3; ++rb) {
        // LDS returns in order: at most 3 younger reads may still be in flight when a[rb] is consumed
        if (s < 3 || rb == 0) LDS_WAIT(3);
        else if (rb == 1) LDS_WAIT(2);
        else if (rb == 2) LDS_WAIT(1);
        else LDS_WAIT(0);
        asm volatile("" : "+v"(a[rb]));   // uses of a[rb] stay below the wait
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rb], bq[s][cb], acc[rb][cb], 0, 0, 0);
        // refill a[rb] in place for the next k-step (the MFMAs above have read it at issue)
        if (s == 0) { if (rb == 0) LDS_RD(a[0], 4128); if (rb == 1) LDS_RD(a[1], 4128 + 512); if (rb == 2) LDS_RD(a[2], 4128 + 1024); if (rb == 3) LDS_RD(a[3], 4128 + 1536); }
        if (s == 1) { if (rb == 0) LDS_RD(a[0], 8256); if (rb == 1) LDS_RD(a[1], 8256 + 512); if (rb == 2) LDS_RD(a[2], 8256 + 1024); if (rb == 3) LDS_RD(a[3], 8256 + 1536); }
        if (s == 2) { if (rb == 0) LDS_RD(a[0], 12384); if (rb == 1) LDS_RD(a[1], 12384 + 512); if (rb == 2) LDS_RD(a[2], 12384 + 1024); if (rb == 3) LDS_RD(a[3], 12384 + 1536); }
      }
      if (!last) {
        const unsigned ksn = (unsigned)((c + 1) * 4 + s) * 1024u;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) bq[s][cb] = ldbuf_bf16x8(rs_w, lane16, w0 + (unsigned)cb * KS * 1024u + ksn);
      }
      if (s == 0) DIAG_STAMP2(c, 2, wave < 4);
    }
#undef LDS_WAIT
#undef LDS_RD
  };
  // Table-row units: set A serves every chunk (depth 1) or the odd chunks (depth 2), set B the even chunks of
  // the depth-2 pipeline.  Depth 2 = rows requested two chunks ahead; the message kernel has the registers for
  // it (and a chunk period shorter than an L2 round trip under load), the coordinate kernel does not.
  constexpr bool DEPTH2 = (CB == 1);   // the message kernel and narrow coordinate MLPs
  UnitH ua0, ua1, ub0, ub1;
  auto vload = [&](UnitH& x0, UnitH& x1, const int cq) {   // table rows for the activations of chunk cq (clamped)
    const int c = cq < NC ? cq : NC - 1;                    // past the end: a harmless repeat of the last chunk
    const unsigned kb = (unsigned)c * kKC3 * 2u;
    unith_load(x0, rs_tab, vdst0, vsrc0, offP + kb, offQ + kb);
    unith_load(x1, rs_tab, vdst1, vsrc1, offP + kb, offQ + kb);
  };
  auto vfinish = [&](const UnitH& x0, const UnitH& x1, const int c) {  // SiLU + bf16 pack of chunk c into its LDS buffer
    const size_t nbuf = (size_t)(c & 1) * kA1_3;
    // vector work wins issue arbitration over the partner wave's MFMAs (which only need 1 slot in 4)
    __builtin_amdgcn_s_setprio(3);
    const bf16x8 o0 = unith_finish(x0, s_wd + c * kKC3 + kg * 8, d2r0, slot0 + nbuf);
    DIAG_STAMP2(c - 1, 1, wave >= 4);
    const bf16x8 o1 = unith_finish(x1, s_wd + c * kKC3 + kg * 8, d2r1, slot1 + nbuf);
    DIAG_STAMP2(c - 1, 2, wave >= 4);
    __builtin_amdgcn_s_setprio(0);
    if constexpr (BWD || SAVE) s1_store(o0, o1, c);
  };
  DIAG_STAMP(30, 2);   // chunk 0 built, first weights requested
  DIAG_RSTAMP(31, 1);
  // The two waves that share a SIMD (w and w+4) run the chunk in opposite phase: waves 0-3 multiply chunk c
  // and then build chunk c+1, waves 4-7 build chunk c+1 first and then multiply chunk c -- one wave's vector
  // work runs under its partner's matrix work instead of both alternating in lockstep.  One barrier per chunk
  // either way.  A unit set is reloaded as soon as it has been consumed (for waves 0-3 that is ahead of the
  // barrier: the vector-memory issue time -- 1 KiB per instruction through a 64 B/clk path -- then overlaps
  // the group's barrier wait instead of delaying its matrix phase).
  // (the steady-state loop bodies are branch-free so that hipcc's waitcnt insertion can keep counted
  // vmcnt waits across the back edge instead of draining the queue at every control-flow join)
  if (wave < 4) {
    auto step = [&](UnitH& x0, UnitH& x1, const int i) {   // chunk i: multiply, build i+1, request the set's next chunk
      DIAG_STAMP(i, 0);
      DIAG_STAMP2(i, 1, true);
      mphase(i, false);
      DIAG_STAMP1(i, 1);
      DIAG_STAMP2(i, 3, true);
      vfinish(x0, x1, i + 1);
      vload(x0, x1, i + (DEPTH2 ? 3 : 2));
      DIAG_STAMP1(i, 2);
      __syncthreads();
      DIAG_STAMP1(i, 3);
    };
    vload(ua0, ua1, 1);
    if constexpr (DEPTH2) {
      vload(ub0, ub1, 2);
      int i = 0;
      for (; i + 1 <= NC - 2; i += 2) { step(ua0, ua1, i); step(ub0, ub1, i + 1); }
      if (i <= NC - 2) step(ua0, ua1, i);
    } else {
      for (int i = 0; i < NC - 1; ++i) step(ua0, ua1, i);
    }
    mphase(NC - 1, true);
    __syncthreads();
  } else {
    auto step = [&](UnitH& x0, UnitH& x1, const int i) {   // build i+1, request the set's next chunk, multiply chunk i
      DIAG_STAMP(i, 0);
      vfinish(x0, x1, i + 1);
      vload(x0, x1, i + (DEPTH2 ? 3 : 2));
      __builtin_amdgcn_sched_barrier(0);
      DIAG_STAMP1(i, 1);
      DIAG_STAMP2(i, 3, true);
      mphase(i, false);
      DIAG_STAMP1(i, 2);
      __syncthreads();
      DIAG_STAMP1(i, 3);
    };
    vload(ua0, ua1, 1);
    if constexpr (DEPTH2) {
      vload(ub0, ub1, 2);
      int i = 0;
      for (; i + 1 <= NC - 2; i += 2) { step(ua0, ua1, i); step(ub0, ub1, i + 1); }
      if (i <= NC - 2) step(ua0, ua1, i);
    } else {
      for (int i = 0; i < NC - 1; ++i) step(ua0, ua1, i);
    }
    mphase(NC - 1, true);
    __syncthreads();
  }
  DIAG_STAMP(30, 3);   // K loop done
  DIAG_RSTAMP(31, 2);

  // row of value index q (q = rb*16 + reg) for this lane
  auto row_of = [&](int q) { return 32 * (q >> 4) + acc_row(q & 15, lane); };

  if constexpr (BWD && !IS_M) {
    // ---- backward of the mlp_x head (:62-65): s = w3 . SiLU(a2) + b3, dL/da2[e][n] = dL/ds_e * w3[n] * SiLU'(a2[e][n]) ----
    const float* s_gsc = s_gseg;
    float part[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) part[q] = 0.f;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int n = 32 * (colblk0 + cb) + r;
      const float bb = p.b2x[n], w3n = p.w3x[n] * kNegLog2e;   // packed vectors carry the -log2(e) / -1/log2(e) scales
      float cs_b = 0.f, cs_w = 0.f;
#pragma unroll
      for (int rb = 0; rb < kRB3; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float sv, ds;
          silu_grad_s(fmaf(acc[rb][cb][i], kNegLog2e, bb), sv, ds);
          part[rb * 16 + i] = fmaf(w3n, sv, part[rb * 16 + i]);
          const float gsc = s_gsc[32 * rb + acc_row(i, lane)];
          const float g = gsc * w3n * ds;
          cs_b += g;
          cs_w = fmaf(gsc, sv, cs_w);
          acc[rb][cb][i] = g;
        }
      cs_b += __shfl_xor(cs_b, 32);
      cs_w += __shfl_xor(cs_w, 32);
      if (hh == 0) { atomicAdd(p.g_col_a + n, cs_b); atomicAdd(p.g_col_b + n, cs_w); }   // g_b2x, g_w3
    }
    {
      float lo[32], hi[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) { lo[q] = part[q]; hi[q] = part[32 + q]; }
      const float t0 = butterfly32(lo, lane), t1 = butterfly32(hi, lane);
      s_part[wave * kR3 + row_of(r)] = t0;
      s_part[wave * kR3 + 64 + row_of(r)] = t1;
    }
    __syncthreads();   // also: every wave is done with the K-loop buffers (reused as store staging below)
    if (tid < kR3) {
      float v = half == 0 ? p.scal[0] : 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += s_part[w * kR3 + tid];
      if (tid < nvalid) p.s_half_out[(size_t)half * p.E + e0 + tid] = v;
    }
    if (half == 0 && wave == 2) {   // g_b3 = sum over edges of dL/ds
      float v = s_gsc[lane] + s_gsc[lane + 64];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if (lane == 0) atomicAdd(p.g_scalar, v);
    }
    // dL/da2 as row-major bf16, 16-byte stores through a per-wave LDS transpose
    __bf16* stg = reinterpret_cast<__bf16*>(s_a1) + (size_t)wave * 32 * 72;
    __bf16* gout = static_cast<__bf16*>(p.g_a2_out) + (size_t)e0 * p.WxP + 32 * colblk0;
#pragma unroll
    for (int rb = 0; rb < kRB3; ++rb) {
      f32x16 blk[2];
      blk[0] = acc[rb][0];
      blk[1] = acc[rb][CB - 1];
      store_block_bf16(blk, CB, stg, gout + (size_t)(32 * rb) * p.WxP, (size_t)p.WxP, nvalid - 32 * rb, lane);
    }
  } else if constexpr (!IS_M) {
    // ---- mlp_x epilogue: s[row] = [b3] + sum_n w3[n] * SiLU(acc + b2[n]) over this workgroup's columns ----
    float part[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) part[q] = 0.f;
    if constexpr (SAVE) {   // the scaled pre-activations go to HBM first (the K-loop buffers are free: every wave passed
                            // the barrier behind the last matrix phase), the accumulators then hold them for the SiLU
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const float bb = p.b2x[32 * (colblk0 + cb) + r];
#pragma unroll
        for (int rb = 0; rb < kRB3; ++rb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[rb][cb][i] = fmaf(acc[rb][cb][i], kNegLog2e, bb);
      }
      __bf16* stg = reinterpret_cast<__bf16*>(s_a1) + (size_t)wave * 32 * 72;
      __bf16* tout = static_cast<__bf16*>(p.g_a2_out) + (size_t)e0 * p.WxP + 32 * colblk0;
#pragma unroll
      for (int rb = 0; rb < kRB3; ++rb) {
        f32x16 blk[2];
        blk[0] = acc[rb][0];
        blk[1] = acc[rb][CB - 1];
        store_block_bf16(blk, CB, stg, tout + (size_t)(32 * rb) * p.WxP, (size_t)p.WxP, nvalid - 32 * rb, lane);
      }
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int n = 32 * (colblk0 + cb) + r;
      const float bb = p.b2x[n], w = p.w3x[n];
#pragma unroll
      for (int rb = 0; rb < kRB3; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          part[rb * 16 + i] = fmaf(w, silu_s(SAVE ? acc[rb][cb][i] : fmaf(acc[rb][cb][i], kNegLog2e, bb)), part[rb * 16 + i]);
    }
    {
      float lo[32], hi[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) { lo[q] = part[q]; hi[q] = part[32 + q]; }
      const float t0 = butterfly32(lo, lane), t1 = butterfly32(hi, lane);
      s_part[wave * kR3 + row_of(r)] = t0;
      s_part[wave * kR3 + 64 + row_of(r)] = t1;
    }
    __syncthreads();
    if (tid < kR3) {
      float v = half == 0 ? p.scal[0] : 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += s_part[w * kR3 + tid];
      s_val[tid] = v;
      if constexpr (SAVE) { if (tid < nvalid) p.s_half_out[(size_t)half * p.E + e0 + tid] = v; }   // share of s_e (backward: g_diff)
    }
    __syncthreads();
    coordinate_segment_sums(p, L, S, tile, half, tid, lane, wave);
  } else {
    static_assert(!IS_M, "edge_bf16_v3.hip keeps the coordinate kernels only (message kernels: edge_bf16_v4.hip)");
  }
  DIAG_STAMP(31, 0);   // epilogue done
}

template <int CB, bool IS_M, bool BWD = false, bool SAVE = false>
int launch_v3(const EdgeParams& p, int blocks, size_t smem, hipStream_t st) {
  hipLaunchKernelGGL((edge_kernel_bf16_v3<CB, IS_M, BWD, SAVE>), dim3(blocks), dim3(kT3), smem, st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace

int init_edge_bf16_v3_attributes() {
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v3<1, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v3<2, false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v3<1, false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v3<1, false, false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return EGNN_OK;
}

int edge_v3_rows() { return kR3; }

bool edge_bf16_v3_supported(const EdgeParams& p) {
  return (p.WxP == 256 || p.WxP == 512 || p.WxP == 1024) && p.MP == 256 && p.WmP % 64 == 0 &&
         v3_smem_bytes(p.WmP, p.MP, true) <= 160 * 1024 && v3_smem_bytes(p.WxP, p.MP, false) <= 160 * 1024 &&
         (size_t)p.N * p.TC * 4 < ((size_t)1 << 32);
}

// forward coordinate kernel for hidden width 256 (one 256-column workgroup per tile); wider coordinate MLPs run on
// edge_x_m16.hip (v_mfma_f32_16x16x32_bf16: 5-6 % faster by wall at the same workgroup tile, profiles/r03c_ab_xm16.log --
// the 512-column forward / training-forward instantiations of this 32x32x16 kernel were retired for it)
int launch_edge_bf16_v3_x(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  if (p.WxP != 256) { set_error("edge_bf16_v3: forward kernel kept for WxP = 256 only"); return EGNN_EINVAL; }
  return launch_v3<1, false>(p, tiles, v3_smem_bytes(p.WxP, p.MP, false), st);
}

// backward recompute of the coordinate branch over the chunk of edges described by p (p.E edges, p.edge_dst / p.edge_src
// already offset to the chunk)
int launch_edge_bf16_v3_x_bwd(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  static_assert(8 * 32 * 72 * 2 <= 2 * kA1_3 + 1024 * 4, "store staging must fit the K-loop buffers");
  if (p.WxP >= 512) return launch_v3<2, false, true>(p, tiles * (p.WxP / 512), v3_smem_bytes(p.WxP, p.MP, false), st);
  return launch_v3<1, false, true>(p, tiles, v3_smem_bytes(p.WxP, p.MP, false), st);
}

// training forward of the coordinate branch for WxP = 256 (wider: edge_x_m16.hip)
int launch_edge_bf16_v3_x_save(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  if (p.WxP != 256) { set_error("edge_bf16_v3: training-forward kernel kept for WxP = 256 only"); return EGNN_EINVAL; }
  return launch_v3<1, false, false, true>(p, tiles, v3_smem_bytes(p.WxP, p.MP, false), st);
}

}  // namespace egnn
