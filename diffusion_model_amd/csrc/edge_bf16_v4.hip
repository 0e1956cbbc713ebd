// Fused per-edge kernels, bf16 MFMA, 128-edge tiles with the output columns split over workgroups (gfx950) -- v4.
//
// Same tiling, prologue and epilogue as edge_bf16_v3.hip; the K loop differs:
//   * v3 runs the two waves of a SIMD in opposite phase (multiply chunk c | build chunk c+1): the partner's vector work
//     hides under this wave's matrix work, but every wave still executes its OWN matrix phase and vector phase one after
//     the other, so a chunk costs (matrix 1.3 k + vector 1.0 k cycles) per wave however well the partners overlap.
//   * v4 interleaves inside each wave: between two groups of MFMAs (which only occupy the issue port for 8 of their 32
//     cycles) the wave finishes ONE activation of the chunk two ahead (SiLU(P[dst] + Q[src] + wd*d2) -> bf16), so its
//     vector work rides in its own MFMA shadow.  The activation chunks live in a 3-deep LDS ring: chunk c is multiplied
//     while chunk c+2 is written, and chunk c+1 (complete since the previous barrier) can already be read for the first
//     k-step of the next iteration BEFORE this iteration's barrier -- no LDS-latency bubble after the barrier.
//   * table rows are re-requested the moment their registers are free (row 0 mid-chunk, row 1 at the end) and have a
//     whole chunk to arrive.
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

#include "kernels.h"

namespace egnn {

namespace {

constexpr int kT3 = 512;
constexpr int kR3 = 128, kRB3 = 4, kRPAD3 = kR3 + 1;
constexpr int kKC3 = 64;
constexpr size_t kA1_3 = (size_t)8 * kRPAD3 * 16;  // one activation chunk [8 k-groups][129][8 bf16]
constexpr int kSegFast3 = 8;

// LDS carve (bytes)
constexpr size_t kOffDst = 0;                                  // int[R]
constexpr size_t kOffSrc = kOffDst + kR3 * 4;                  // int[R]
constexpr size_t kOffD2 = kOffSrc + kR3 * 4;                   // float[R]
constexpr size_t kOffDiff = kOffD2 + kR3 * 4;                  // float[3][R]
constexpr size_t kOffVal = kOffDiff + 3 * kR3 * 4;             // float[R]   s_ij (X) / gate (M)
constexpr size_t kOffPart = kOffVal + kR3 * 4;                 // float[8][R] per-wave partial row sums
constexpr size_t kOffSegRow = kOffPart + 8 * kR3 * 4;          // int[R]
constexpr size_t kOffSegNode = kOffSegRow + kR3 * 4;           // int[R]
constexpr size_t kOffSegRs = kOffSegNode + kR3 * 4;            // int[R]
constexpr size_t kOffSegRe = kOffSegRs + kR3 * 4;              // int[R]
constexpr size_t kOffSegMode = kOffSegRe + kR3 * 4;            // int[R]
constexpr size_t kOffMisc = kOffSegMode + kR3 * 4;             // int[16]
constexpr size_t kOffGseg = kOffMisc + 64;                     // float[kSegFast3][R]
constexpr size_t kOffA1 = kOffGseg + kSegFast3 * kR3 * 4;      // ring of 3 activation chunks, then wd[KP]
constexpr int kRing = 3;
__host__ __device__ inline size_t v4_smem_bytes(int KP, int MP, bool is_m) {
  (void)MP; (void)is_m;
  return kOffA1 + kRing * kA1_3 + (size_t)KP * 4;
}

#ifndef EGNN_V4_M_WAVES
#define EGNN_V4_M_WAVES 4   // waves per SIMD the message kernel is compiled for (4 = two workgroups per CU)
#endif
// BWD = true (message kernel): the training backward's recompute pass over a chunk of edges: the activation chunks are
// also written to HBM (s1_out) and the epilogue produces dL/d(a2m) through the gate instead of the segment sums.
// SAVE = true (message kernel): the training forward: also leaves the activation chunks (s1_out) and the scaled
// second-layer pre-activations (g_a2_out) in HBM (see edge_bf16_v3.hip).
template <int CB, bool IS_M, bool BWD = false, bool SAVE = false>
__global__ __launch_bounds__(kT3, ((CB == 1 && !BWD) ? EGNN_V4_M_WAVES : 2)) void edge_kernel_bf16_v4(const EdgeParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_dst = reinterpret_cast<int*>(smem + kOffDst);
  int* s_src = reinterpret_cast<int*>(smem + kOffSrc);
  float* s_d2 = reinterpret_cast<float*>(smem + kOffD2);
  float* s_diff = reinterpret_cast<float*>(smem + kOffDiff);
  float* s_val = reinterpret_cast<float*>(smem + kOffVal);
  float* s_part = reinterpret_cast<float*>(smem + kOffPart);
  int* s_seg_of_row = reinterpret_cast<int*>(smem + kOffSegRow);
  int* s_seg_node = reinterpret_cast<int*>(smem + kOffSegNode);
  int* s_seg_rs = reinterpret_cast<int*>(smem + kOffSegRs);
  int* s_seg_re = reinterpret_cast<int*>(smem + kOffSegRe);
  int* s_seg_mode = reinterpret_cast<int*>(smem + kOffSegMode);
  int* s_misc = reinterpret_cast<int*>(smem + kOffMisc);
  float* s_gseg = reinterpret_cast<float*>(smem + kOffGseg);
  char* s_a1 = smem + kOffA1;
  float* s_wd = reinterpret_cast<float*>(s_a1 + kRing * kA1_3);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int KP = IS_M ? p.WmP : p.WxP;
  const int nsplit = IS_M ? 1 : p.WxP / (256 * CB);
  const int j = xcd_tile(blockIdx.x, gridDim.x);
  const int tile = j / nsplit, half = j - tile * nsplit;
  const int e0 = tile * kR3;
  const int nvalid = min(kR3, p.E - e0);

#ifdef EGNN_EXP_STAMP
  const bool stamp_wg = blockIdx.x == gridDim.x / 2;
  unsigned long long* st_base = p.stamps + ((size_t)(IS_M ? 1 : 0) * 8 + wave) * 32 * 4;
#define STAMP(c, k)                                                                               \
  do {                                                                                            \
    unsigned long long t_;                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (stamp_wg && lane == 0 && (c) < 32) st_base[(c) * 4 + (k)] = t_;                           \
  } while (0)
// 100 MHz wall counter next to a cycle stamp: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
#define RSTAMP(c, k)                                                                              \
  do {                                                                                            \
    unsigned long long t_;                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (stamp_wg && lane == 0 && (c) < 32) st_base[(c) * 4 + (k)] = t_;                           \
  } while (0)
#else
#define STAMP(c, k)
#define RSTAMP(c, k)
#endif
#ifdef EGNN_EXP_STAMP2   // finer stamps inside the first half of a chunk (replaces the meaning of slots 1..3)
#define STAMP2(c, k, cond) do { if (cond) STAMP(c, k); } while (0)
#define STAMP1(c, k)
#else
#define STAMP2(c, k, cond)
#define STAMP1(c, k) STAMP(c, k)
#endif
  STAMP(30, 0);   // kernel entry
  // ---- prologue: edge rows, geometry, segment (= receiving node) structure ----
  if (tid < kR3) {
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
    s_diff[tid] = dx; s_diff[kR3 + tid] = dy; s_diff[2 * kR3 + tid] = dz;
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);  // norm(...)**2 as in the reference (:56)
    s_d2[tid] = nrm * nrm;
  }
  {
    const float* wd = IS_M ? p.wdm : p.wdx;
    for (int i = tid; i < KP; i += kT3) s_wd[i] = wd[i];
  }
  __syncthreads();
  bool is_start = false, is_end = false;
  unsigned long long starts = 0;
  if (tid < kR3) {   // waves 0 and 1
    const bool valid = tid < nvalid;
    const int d = s_dst[tid];
    is_start = valid && (tid == 0 || s_dst[tid - 1] != d);
    is_end = valid && (tid == nvalid - 1 || s_dst[tid + 1] != d);
    starts = __ballot(is_start);
    if (lane == 0) s_misc[1 + wave] = __popcll(starts);
  }
  __syncthreads();
  if (tid < kR3) {
    const int seg = (wave == 1 ? s_misc[1] : 0) + __popcll(starts & ((2ull << lane) - 1ull)) - 1;
    s_seg_of_row[tid] = tid < nvalid ? seg : -1;
    if (is_start) { s_seg_node[seg] = s_dst[tid]; s_seg_rs[seg] = tid; }
    if (is_end) s_seg_re[seg] = tid;
    if (tid == 0) s_misc[0] = s_misc[1] + s_misc[2];
  }
  __syncthreads();
  const int S = s_misc[0];
  if (tid < S) {   // where does each segment's sum go?  (same rule as the other edge kernels)
    const int n = s_seg_node[tid];
    const bool first = (e0 + s_seg_rs[tid]) == p.row_ptr[n];
    const bool last = (e0 + s_seg_re[tid] + 1) == p.row_ptr[n + 1];
    s_seg_mode[tid] = (first && last) ? 2 : (first ? 1 : 0);
  }

  STAMP(30, 1);   // tile structure ready

  // ---- K-loop ----
  const int NC = KP / kKC3, KS = KP / 16;
  const int brow = tid >> 3, kg = tid & 7;   // this thread builds rows brow and brow + 64, columns [8 kg, 8 kg + 8) of a chunk
  const rsrc_t rs_tab = make_rsrc(p.table, (p.dbg & 2) ? 0u : (unsigned)((size_t)p.N * p.TC * 2));
  const rsrc_t rs_w = make_rsrc(IS_M ? p.w2m : p.w2x, (p.dbg & 1) ? 0u : (unsigned)((size_t)(IS_M ? p.MP : p.WxP) * KP * 2));
  const unsigned vdst0 = (unsigned)s_dst[brow] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const unsigned vsrc0 = (unsigned)s_src[brow] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const unsigned vdst1 = (unsigned)s_dst[brow + 64] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const unsigned vsrc1 = (unsigned)s_src[brow + 64] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const float d2r0 = s_d2[brow], d2r1 = s_d2[brow + 64];
  const unsigned offP = (IS_M ? 2u * p.WxP : 0u) * 2u, offQ = (IS_M ? 2u * p.WxP + p.WmP : (unsigned)p.WxP) * 2u;   // fp16 table
  char* slot0 = s_a1 + ((size_t)kg * kRPAD3 + brow) * 16;
  char* slot1 = slot0 + 64 * 16;
  const unsigned lane16 = lane * 16u;
  // 32-bit LDS byte address of this lane's A-fragment slot in ring buffer 0
  const unsigned lds_a1_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_a1 + ((size_t)hh * kRPAD3 + r) * 16);
  const int colblk0 = half * 8 * CB + wave * CB;   // first 32-column block of this wave
  const unsigned w0off = (unsigned)colblk0 * KS * 1024u;

  f32x16 acc[kRB3][CB];
#pragma unroll
  for (int rb = 0; rb < kRB3; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;

  UnitH u0, u1;   // table rows of rows brow / brow + 64 for the chunk that is built next
  auto uload = [&](UnitH& u, const unsigned vd, const unsigned vs, const int cq) {
    const int c = cq < NC ? cq : NC - 1;                    // past the end: a harmless repeat of the last chunk
    const unsigned kb = (unsigned)c * kKC3 * 2u;
    unith_load(u, rs_tab, vd, vs, offP + kb, offQ + kb);
  };
  // chunks 0 and 1 are built up front (ring slots 0 and 1)
  // backward: the activation chunk also goes to HBM, 16 bytes per thread, 128 contiguous bytes per row and chunk
  auto s1_store = [&](const bf16x8 ov, const int row, const int c) {
    if (half != 0) return;
    if (row < nvalid)
      *reinterpret_cast<bf16x8*>(static_cast<__bf16*>(p.s1_out) + (size_t)(e0 + row) * KP + c * kKC3 + kg * 8) = ov;
  };
  uload(u0, vdst0, vsrc0, 0); uload(u1, vdst1, vsrc1, 0);
  {
    const bf16x8 o0 = unith_finish(u0, s_wd + kg * 8, d2r0, slot0);
    const bf16x8 o1 = unith_finish(u1, s_wd + kg * 8, d2r1, slot1);
    if constexpr (BWD || SAVE) { s1_store(o0, brow, 0); s1_store(o1, brow + 64, 0); }
  }
  uload(u0, vdst0, vsrc0, 1); uload(u1, vdst1, vsrc1, 1);
  {
    const bf16x8 o0 = unith_finish(u0, s_wd + kKC3 + kg * 8, d2r0, slot0 + kA1_3);
    const bf16x8 o1 = unith_finish(u1, s_wd + kKC3 + kg * 8, d2r1, slot1 + kA1_3);
    if constexpr (BWD || SAVE) { s1_store(o0, brow, 1); s1_store(o1, brow + 64, 1); }
  }
  uload(u0, vdst0, vsrc0, 2); uload(u1, vdst1, vsrc1, 2);
  // weight fragments, requested BQD k-steps ahead of their use (a whole chunk for the coordinate kernel; the message
  // kernel, compiled for <= 128 VGPRs so that two workgroups share a CU, keeps 2 -- its other workgroup covers the rest)
  constexpr int BQD = (CB == 1 && !BWD && EGNN_V4_M_WAVES >= 4) ? 2 : 4;
  bf16x8 bq[BQD][CB];
#pragma unroll
  for (int s = 0; s < BQD; ++s)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) bq[s][cb] = ldbuf_bf16x8(rs_w, lane16, w0off + ((unsigned)cb * KS + s) * 1024u);
  __syncthreads();

  // ring offsets (bytes): chunk c is read at off_cur, chunk c+1 at off_nxt, chunk c+2 is written at off_wr
  unsigned off_cur = 0u, off_nxt = (unsigned)kA1_3, off_wr = 2u * (unsigned)kA1_3;
  bf16x8 a[kRB3];
#define LDS_RD(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define LDS_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
  LDS_RD(a[0], lds_a1_base, 0); LDS_RD(a[1], lds_a1_base, 512); LDS_RD(a[2], lds_a1_base, 1024); LDS_RD(a[3], lds_a1_base, 1536);

  // One chunk.  16 groups (k-step s, row block rb): wait for a[rb] (LDS returns in order: the 3 younger refills may
  // still fly), CB MFMAs, refill a[rb] in place for the next k-step (from the NEXT chunk's buffer after the last
  // k-step), and ONE activation of chunk c+2 in two halves of ~16 issue cycles, each behind an MFMA: an in-order wave can
  // only use the 24 free issue cycles of an MFMA's 32 if the fillers sit BETWEEN two MFMAs in program order.
  // sched_barrier(0) pins this order; inside a half the compiler schedules freely.
  auto chunk = [&](const int c, const bool build, const bool last) {
    STAMP(c, 0);
    const unsigned abase = lds_a1_base + off_cur, nbase = lds_a1_base + off_nxt;
    f16x8 t;
    bf16x8 o;
    const float* wdc = s_wd + (c + 2) * kKC3 + kg * 8;   // d^2 column of the first layer for this thread's 8 hidden units
    // The 16 activations a thread owes to chunk c+2 are finished 4 per k-step, as a software pipeline over the 8 MFMA
    // gaps of the k-step: every gap issues ~16 cycles of INDEPENDENT vector instructions whose operands were produced at
    // least one gap earlier (an in-order wave stalls on a dependent transcendental otherwise), between two MFMAs in
    // program order (that is where an in-order wave can use the 24 issue cycles an MFMA leaves free of its 32):
    //   gap 0: 4 x fma_mix (pre-activation)   1-2: 2 x exp2 each   3: 4 x add 1   4-5: 2 x rcp each   6: 4 x mul
    //   gap 7: bf16 pack; every second k-step completes a 16-byte LDS slot and re-requests the consumed table row.
    float pu[4], pe[4];
#define STAGE(S, Q)                                                                                           \
    if (build) {                                                                                              \
      constexpr int e0_ = 4 * ((S) & 1);   /* first element of this k-step inside its unit */                 \
      if ((Q) == 0) {                                                                                         \
        if ((S) == 0) t = u0.p + u0.q;                                                                        \
        if ((S) == 2) t = u1.p + u1.q;                                                                        \
        const f32x4 wv_ = *reinterpret_cast<const f32x4*>(wdc + e0_);                                         \
        _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                         \
          pu[k] = fmaf(wv_[k], (S) < 2 ? d2r0 : d2r1, (float)t[e0_ + k]);                                     \
      }                                                                                                       \
      if ((Q) == 1) { pe[0] = __builtin_amdgcn_exp2f(pu[0]); pe[1] = __builtin_amdgcn_exp2f(pu[1]); }         \
      if ((Q) == 2) { pe[2] = __builtin_amdgcn_exp2f(pu[2]); pe[3] = __builtin_amdgcn_exp2f(pu[3]); }         \
      if ((Q) == 3) { _Pragma("unroll") for (int k = 0; k < 4; ++k) pe[k] = 1.0f + pe[k]; }                   \
      if ((Q) == 4) { pe[0] = __builtin_amdgcn_rcpf(pe[0]); pe[1] = __builtin_amdgcn_rcpf(pe[1]); }           \
      if ((Q) == 5) { pe[2] = __builtin_amdgcn_rcpf(pe[2]); pe[3] = __builtin_amdgcn_rcpf(pe[3]); }           \
      if ((Q) == 6) { _Pragma("unroll") for (int k = 0; k < 4; ++k) pu[k] = pu[k] * pe[k]; }                  \
      if ((Q) == 7) {                                                                                         \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) o[e0_ + k] = (__bf16)pu[k];                             \
        if ((S) == 1) { *reinterpret_cast<bf16x8*>(slot0 + off_wr) = o; uload(u0, vdst0, vsrc0, c + 3);       \
                        if constexpr (BWD || SAVE) s1_store(o, brow, c + 2); }                                        \
        if ((S) == 3) { *reinterpret_cast<bf16x8*>(slot1 + off_wr) = o; uload(u1, vdst1, vsrc1, c + 3);       \
                        if constexpr (BWD || SAVE) s1_store(o, brow + 64, c + 2); }                                   \
      }                                                                                                       \
    }
#define GROUP(S, RB)                                                                                          \
    {                                                                                                         \
      if (!last || (S) < 3 || (RB) == 0) LDS_WAIT(3);                                                         \
      else if ((RB) == 1) LDS_WAIT(2);                                                                        \
      else if ((RB) == 2) LDS_WAIT(1);                                                                        \
      else LDS_WAIT(0);                                                                                       \
      asm volatile("" : "+v"(a[RB]));                                                                         \
      acc[RB][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[RB], bq[(S) % BQD][0], acc[RB][0], 0, 0, 0);     \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      STAGE(S, 2 * (RB))                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      _Pragma("unroll") for (int cb = 1; cb < CB; ++cb)                                                       \
        acc[RB][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[RB], bq[(S) % BQD][cb], acc[RB][cb], 0, 0, 0); \
      if ((S) < 3) LDS_RD(a[RB], abase, ((S) + 1) * 4128 + (RB) * 512);                                       \
      else if (!last) LDS_RD(a[RB], nbase, (RB) * 512);                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      STAGE(S, 2 * (RB) + 1)                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
    }
#ifdef EGNN_V4_PRIO_FLIP   /* alternate which of the two waves of a SIMD wins arbitration, per k-step */
#define KPRIO(S) if (((((S) & 1) ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#else
#define KPRIO(S)
#endif
#define KSTEP(S)                                                                                              \
    KPRIO(S)                                                                                                  \
    GROUP(S, 0) GROUP(S, 1) GROUP(S, 2) GROUP(S, 3)                                                           \
    if (!last || (S) + BQD < 4) {   /* k-step c*4 + S + BQD: this chunk's or the next one's */                \
      const unsigned ksn = (unsigned)(c * 4 + (S) + BQD) * 1024u;                                             \
      _Pragma("unroll") for (int cb = 0; cb < CB; ++cb)                                                       \
        bq[(S) % BQD][cb] = ldbuf_bf16x8(rs_w, lane16, w0off + (unsigned)cb * KS * 1024u + ksn);              \
    }
    KSTEP(0) KSTEP(1)
    STAMP(c, 1);
    KSTEP(2) KSTEP(3)
    STAMP(c, 2);
#undef KSTEP
#undef GROUP
#undef STAGE
    const unsigned tmp = off_cur; off_cur = off_nxt; off_nxt = off_wr; off_wr = tmp;
  };
  // (the steady-state loop body is branch-free so that hipcc's waitcnt insertion keeps counted vmcnt waits across
  // the back edge instead of draining the queue at every control-flow join)
  STAMP(30, 2);   // chunks 0 and 1 built, first weights requested
  RSTAMP(31, 1);
  for (int c = 0; c < NC - 2; ++c) { chunk(c, true, false); __syncthreads(); STAMP(c, 3); }
  chunk(NC - 2, false, false);
  __syncthreads();
  STAMP(NC - 2, 3);
  chunk(NC - 1, false, true);
  __syncthreads();
  STAMP(NC - 1, 3);
  RSTAMP(31, 2);
#undef LDS_WAIT
#undef LDS_RD
  STAMP(30, 3);   // K loop done


  // row of value index q (q = rb*16 + reg) for this lane
  auto row_of = [&](int q) { return 32 * (q >> 4) + acc_row(q & 15, lane); };

  if constexpr (BWD && IS_M) {
    // ---- backward of the message head (:57-60): m = SiLU(a2), z = wa . m + ba, gate = sigmoid(z), out = m * gate;
    //      with g = dL/d(sum_m[i]):  dL/dm = g * gate + (g . m) gate (1 - gate) wa,  dL/da2 = dL/dm * SiLU'(a2) ----
    static_assert(!(BWD && IS_M) || CB == 1, "message epilogue assumes one 32-column block per wave");
    const int ncol = 32 * wave + r;
    const float bb = p.b2m[ncol], wan = p.wa[ncol] * kNegLog2e;   // packed vectors carry the -log2(e) / -1/log2(e) scales
    const rsrc_t rs_gm = make_rsrc(p.g_sum_m, (unsigned)((size_t)p.N * p.MP * 4));
    const unsigned gcol = 4u * (unsigned)ncol, gld = 4u * (unsigned)p.MP;
    auto gm_at = [&](const int row) {   // dL/d(sum_m)[dst(row)][ncol]
      return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_gm, (unsigned)s_dst[row] * gld + gcol, 0, 0));
    };
    // pass 1: row sums z = wa . m and d = g . m (two halves of 64 rows to bound the live registers)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float vz[32], vd[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int rb = 2 * hf + (q >> 4), i = q & 15, row = 32 * rb + acc_row(i, lane);
        float m, ds;
        silu_grad_s(fmaf(acc[rb][0][i], kNegLog2e, bb), m, ds);
        vz[q] = wan * m;
        vd[q] = gm_at(row) * m;
      }
      const float tz = butterfly32(vz, lane), td = butterfly32(vd, lane);
      s_part[wave * kR3 + 64 * hf + row_of(r)] = tz;
      s_gseg[wave * kR3 + 64 * hf + row_of(r)] = td;
    }
    __syncthreads();
    if (tid < kR3) {
      float z = p.scal[1], d = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { z += s_part[w * kR3 + tid]; d += s_gseg[w * kR3 + tid]; }
      const float gate = sigmoid_f(z);
      const bool valid = tid < nvalid;
      s_val[tid] = gate;
      s_d2[tid] = valid ? d * gate * (1.0f - gate) : 0.f;   // coef (the geometry is no longer needed)
    }
    __syncthreads();
    if (wave == 2) {   // g_ba = sum over edges of coef
      float v = s_d2[lane] + s_d2[lane + 64];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if (lane == 0) atomicAdd(p.g_scalar, v);
    }
    // pass 2: dL/da2, column sums, store
    float cs_b = 0.f, cs_w = 0.f;
#pragma unroll
    for (int rb = 0; rb < kRB3; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = 32 * rb + acc_row(i, lane);
        float m, ds;
        silu_grad_s(fmaf(acc[rb][0][i], kNegLog2e, bb), m, ds);
        const float coef = s_d2[row];
        const float g = row < nvalid ? fmaf(gm_at(row), s_val[row], coef * wan) * ds : 0.f;
        cs_b += g;
        cs_w = fmaf(coef, m, cs_w);
        acc[rb][0][i] = g;
      }
    cs_b += __shfl_xor(cs_b, 32);
    cs_w += __shfl_xor(cs_w, 32);
    if (hh == 0) { atomicAdd(p.g_col_a + ncol, cs_b); atomicAdd(p.g_col_b + ncol, cs_w); }   // g_b2m, g_wa
    __bf16* stg = reinterpret_cast<__bf16*>(s_a1) + (size_t)wave * 32 * 72;
    __bf16* gout = static_cast<__bf16*>(p.g_a2_out) + (size_t)e0 * p.MP + 32 * wave;
#pragma unroll
    for (int rb = 0; rb < kRB3; ++rb) {
      f32x16 blk[2];
      blk[0] = acc[rb][0];
      blk[1] = acc[rb][0];
      store_block_bf16(blk, 1, stg, gout + (size_t)(32 * rb) * p.MP, (size_t)p.MP, nvalid - 32 * rb, lane);
    }
  } else if constexpr (!IS_M) {
    // ---- mlp_x epilogue: s[row] = [b3] + sum_n w3[n] * SiLU(acc + b2[n]) over this workgroup's columns ----
    float part[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) part[q] = 0.f;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int n = 32 * (colblk0 + cb) + r;
      const float bb = p.b2x[n], w = p.w3x[n];
#pragma unroll
      for (int rb = 0; rb < kRB3; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) part[rb * 16 + i] = fmaf(w, silu_s(fmaf(acc[rb][cb][i], kNegLog2e, bb)), part[rb * 16 + i]);
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
    }
    __syncthreads();
    float* aggx = p.agg_x + (size_t)half * p.agg_x_stride;
    float* partx = p.part_x + (size_t)half * p.part_x_stride;
    // Component 3 of every coordinate sum carries the segment's sum of |x_i - x_j|^2 (plain squares: the Frobenius norm
    // of :64 is sqrt of the sum over ALL edges), so the normaliser needs no pass of its own over the edges.
    if (S <= kSegFast3) {
      if (wave == 0) {  // coordinate messages (x_i - x_j) * s_ij; 1/(G+1) is applied in node_post
        float c[2][4];
        int myseg[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int row = lane + 64 * u;
          myseg[u] = s_seg_of_row[row];
          const float sv = s_val[row];
          const float dx = s_diff[row], dy = s_diff[kR3 + row], dz = s_diff[2 * kR3 + row];
          c[u][0] = dx * sv; c[u][1] = dy * sv; c[u][2] = dz * sv;
          c[u][3] = dx * dx + dy * dy + dz * dz;
        }
        for (int seg = 0; seg < S; ++seg) {
          float a0 = (myseg[0] == seg ? c[0][0] : 0.f) + (myseg[1] == seg ? c[1][0] : 0.f);
          float a1 = (myseg[0] == seg ? c[0][1] : 0.f) + (myseg[1] == seg ? c[1][1] : 0.f);
          float a2 = (myseg[0] == seg ? c[0][2] : 0.f) + (myseg[1] == seg ? c[1][2] : 0.f);
          float a3 = (myseg[0] == seg ? c[0][3] : 0.f) + (myseg[1] == seg ? c[1][3] : 0.f);
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) { a0 += __shfl_xor(a0, m); a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); a3 += __shfl_xor(a3, m); }
          if (lane < 4) {
            const int mode = s_seg_mode[seg];
            float* dstp = mode == 2 ? aggx + (size_t)s_seg_node[seg] * 4 : partx + ((size_t)tile * 2 + mode) * 4;
            dstp[lane] = lane == 0 ? a0 : (lane == 1 ? a1 : (lane == 2 ? a2 : a3));
          }
        }
      }
    } else {
      for (int t = tid; t < 4 * S; t += kT3) {
        const int seg = t >> 2, d = t & 3, mode = s_seg_mode[seg];
        float sum = 0.f;
        if (d < 3) {
          for (int rr = s_seg_rs[seg]; rr <= s_seg_re[seg]; ++rr) sum += s_diff[d * kR3 + rr] * s_val[rr];
        } else {
          for (int rr = s_seg_rs[seg]; rr <= s_seg_re[seg]; ++rr) {
            const float dx = s_diff[rr], dy = s_diff[kR3 + rr], dz = s_diff[2 * kR3 + rr];
            sum += dx * dx + dy * dy + dz * dz;
          }
        }
        float* dstp = mode == 2 ? aggx + (size_t)s_seg_node[seg] * 4 : partx + ((size_t)tile * 2 + mode) * 4;
        dstp[d] = sum;
      }
    }
  } else {
    // ---- mlp_m epilogue: m = SiLU(acc + b2), gate = sigmoid(wa . m + ba) (:31-34, :59-60) ----
    static_assert(!IS_M || CB == 1, "message epilogue assumes one 32-column block per wave");
    float mval[64];
    const int ncol = 32 * wave + r;
    if constexpr (SAVE) {   // scaled pre-activations to HBM first (the ring is free behind the last chunk's barrier)
      const float bb = p.b2m[ncol];
#pragma unroll
      for (int rb = 0; rb < kRB3; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rb][0][i] = fmaf(acc[rb][0][i], kNegLog2e, bb);
      __bf16* stg = reinterpret_cast<__bf16*>(s_a1) + (size_t)wave * 32 * 72;
      __bf16* tout = static_cast<__bf16*>(p.g_a2_out) + (size_t)e0 * p.MP + 32 * wave;
#pragma unroll
      for (int rb = 0; rb < kRB3; ++rb) {
        f32x16 blk[2];
        blk[0] = acc[rb][0];
        blk[1] = acc[rb][0];
        store_block_bf16(blk, 1, stg, tout + (size_t)(32 * rb) * p.MP, (size_t)p.MP, nvalid - 32 * rb, lane);
      }
    }
    {
      const float bb = p.b2m[ncol], wa = p.wa[ncol];
      float lo[32], hi[32];
#pragma unroll
      for (int rb = 0; rb < kRB3; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float m = silu_s(SAVE ? acc[rb][0][i] : fmaf(acc[rb][0][i], kNegLog2e, bb));   // = -log2(e) * m
          mval[rb * 16 + i] = m;
          if (rb < 2) lo[rb * 16 + i] = wa * m; else hi[(rb - 2) * 16 + i] = wa * m;
        }
      const float t0 = butterfly32(lo, lane), t1 = butterfly32(hi, lane);
      s_part[wave * kR3 + row_of(r)] = t0;
      s_part[wave * kR3 + 64 + row_of(r)] = t1;
    }
    __syncthreads();
    if (tid < kR3) {
      float g = p.scal[1];
#pragma unroll
      for (int w = 0; w < 8; ++w) g += s_part[w * kR3 + tid];
      s_val[tid] = sigmoid_f(g) * kNegInvLog2e;   // also undoes the scale of mval
    }
    __syncthreads();
    // segment sums, kSegFast3 segments per pass (one pass unless the tile holds many short segments): the gate of
    // each row is laid out per segment in LDS and every lane dots its 64 message values with it
    for (int base = 0; base < S; base += kSegFast3) {
      const int ns = min(kSegFast3, S - base);
      if (base > 0) __syncthreads();   // the previous pass has been read
      for (int t = tid; t < ns * kR3; t += kT3) {
        const int seg = base + (t >> 7), row = t & 127;
        s_gseg[t] = (s_seg_of_row[row] == seg) ? s_val[row] : 0.f;
      }
      __syncthreads();
      for (int sg = 0; sg < ns; ++sg) {
        const int seg = base + sg;
        const float* gw = s_gseg + sg * kR3 + 4 * hh;
        float v = 0.f;
#pragma unroll
        for (int rb = 0; rb < kRB3; ++rb)
#pragma unroll
          for (int i = 0; i < 16; ++i) v = fmaf(mval[rb * 16 + i], gw[32 * rb + (i & 3) + 8 * (i >> 2)], v);
        v += __shfl_xor(v, 32);
        if (hh == 0) {
          const int mode = s_seg_mode[seg];
          float* dstp = mode == 2 ? p.agg_m + (size_t)s_seg_node[seg] * p.MP : p.part_m + ((size_t)tile * 2 + mode) * p.MP;
          dstp[ncol] = v;
        }
      }
    }
  }
  STAMP(31, 0);   // epilogue done
}

template <int CB, bool IS_M, bool BWD = false, bool SAVE = false>
int launch_v4(const EdgeParams& p, int blocks, size_t smem, hipStream_t st) {
  hipLaunchKernelGGL((edge_kernel_bf16_v4<CB, IS_M, BWD, SAVE>), dim3(blocks), dim3(kT3), smem, st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace

int init_edge_bf16_v4_attributes() {
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v4<2, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v4<1, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v4<1, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v4<1, true, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v4<1, true, false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return EGNN_OK;
}

int edge_v4_rows() { return kR3; }

bool edge_bf16_v4_supported(const EdgeParams& p) {
  return (p.WxP == 256 || p.WxP == 512 || p.WxP == 1024) && p.MP == 256 && p.WmP % 64 == 0 &&
         v4_smem_bytes(p.WmP, p.MP, true) <= 160 * 1024 && v4_smem_bytes(p.WxP, p.MP, false) <= 160 * 1024 &&
         (size_t)p.N * p.TC * 4 < ((size_t)1 << 32);
}

// backward recompute of the message branch over the chunk of edges described by p
int launch_edge_bf16_v4_m_bwd(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  static_assert(8 * 32 * 72 * 2 <= kRing * kA1_3, "store staging must fit the K-loop buffers");
  return launch_v4<1, true, true>(p, tiles, v4_smem_bytes(p.WmP, p.MP, true), st);
}

// coordinate kernel as 256-column workgroups (two per CU), WxP / 256 column shares per tile
int launch_edge_bf16_v4_x1(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  return launch_v4<1, false>(p, tiles * (p.WxP / 256), v4_smem_bytes(p.WxP, p.MP, false), st);
}

// training forward of the message branch: p.s1_out / p.g_a2_out receive the activations and the scaled pre-activations
int launch_edge_bf16_v4_m_save(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  return launch_v4<1, true, false, true>(p, tiles, v4_smem_bytes(p.WmP, p.MP, true), st);
}

// message kernel only
int launch_edge_bf16_v4_m(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  return launch_v4<1, true>(p, tiles, v4_smem_bytes(p.WmP, p.MP, true), st);
}

int launch_edge_bf16_v4(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  int rc;
  // X: CB = 2 (512 columns per workgroup) when the hidden width allows, else one 256-column workgroup
  if (p.WxP >= 512) rc = launch_v4<2, false>(p, tiles * (p.WxP / 512), v4_smem_bytes(p.WxP, p.MP, false), st);
  else rc = launch_v4<1, false>(p, tiles, v4_smem_bytes(p.WxP, p.MP, false), st);
  if (rc) return rc;
  return launch_v4<1, true>(p, tiles, v4_smem_bytes(p.WmP, p.MP, true), st);
}

}  // namespace egnn
