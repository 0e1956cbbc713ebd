// Coordinate edge kernel, 16 wave64 per workgroup (gfx950): the 128-edge x 512-column tile of edge_bf16_v3/v4.hip held by
// SIXTEEN waves of 4 x 1 accumulator tiles (64 accumulator registers each, <= 128 VGPRs per wave) instead of eight waves
// of 4 x 2: four waves per SIMD instead of two.  The activations of mlp_x are still built once per workgroup (twice per
// tile, as before) -- one 8-column unit per thread and chunk -- but four waves per SIMD give the scheduler something to
// issue while one wave waits on the transcendental port, the MFMA pipe, LDS or the barrier; the two-wave kernels lose a
// third of every chunk to exactly those waits (profiles/r02_stamps_*.txt).  K loop = the in-wave pipeline of
// edge_bf16_v4.hip (3-deep LDS ring, activations finished in the MFMA gaps).
#include <type_traits>

#include "kernels.h"

namespace egnn {

namespace {

constexpr int kT3 = 1024, kNW = 16;
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
constexpr size_t kOffPart = kOffVal + kR3 * 4;                 // float[16][R] per-wave partial row sums
constexpr size_t kOffSegRow = kOffPart + kNW * kR3 * 4;        // int[R]
constexpr size_t kOffSegNode = kOffSegRow + kR3 * 4;           // int[R]
constexpr size_t kOffSegRs = kOffSegNode + kR3 * 4;            // int[R]
constexpr size_t kOffSegRe = kOffSegRs + kR3 * 4;              // int[R]
constexpr size_t kOffSegMode = kOffSegRe + kR3 * 4;            // int[R]
constexpr size_t kOffMisc = kOffSegMode + kR3 * 4;             // int[16]
constexpr size_t kOffGseg = kOffMisc + 64;                     // float[kSegFast3][R]
constexpr size_t kOffA1 = kOffGseg + kSegFast3 * kR3 * 4;      // ring of 3 activation chunks, then wd[KP]
constexpr int kRing = 3;
__host__ __device__ inline size_t x16_smem_bytes(int KP, int MP, bool is_m) {
  (void)MP; (void)is_m;
  return kOffA1 + kRing * kA1_3 + (size_t)KP * 4;
}

__global__ __launch_bounds__(kT3, 4) void edge_kernel_bf16_x16(const EdgeParams p) {
  constexpr int CB = 1;
  constexpr bool IS_M = false;
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
  char* s_a1 = smem + kOffA1;
  float* s_wd = reinterpret_cast<float*>(s_a1 + kRing * kA1_3);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int KP = IS_M ? p.WmP : p.WxP;
  const int nsplit = p.WxP / (32 * kNW);
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
  const int brow = tid >> 3, kg = tid & 7;   // this thread builds row brow (0..127), columns [8 kg, 8 kg + 8) of a chunk
  const rsrc_t rs_tab = make_rsrc(p.table, (p.dbg & 2) ? 0u : (unsigned)((size_t)p.N * p.TC * 2));
  const rsrc_t rs_w = make_rsrc(IS_M ? p.w2m : p.w2x, (p.dbg & 1) ? 0u : (unsigned)((size_t)(IS_M ? p.MP : p.WxP) * KP * 2));
  const unsigned vdst0 = (unsigned)s_dst[brow] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const unsigned vsrc0 = (unsigned)s_src[brow] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const float d2r0 = s_d2[brow];
  const unsigned offP = (IS_M ? 2u * p.WxP : 0u) * 2u, offQ = (IS_M ? 2u * p.WxP + p.WmP : (unsigned)p.WxP) * 2u;   // fp16 table
  char* slot0 = s_a1 + ((size_t)kg * kRPAD3 + brow) * 16;
  const unsigned lane16 = lane * 16u;
  // 32-bit LDS byte address of this lane's A-fragment slot in ring buffer 0
  const unsigned lds_a1_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_a1 + ((size_t)hh * kRPAD3 + r) * 16);
  const int colblk0 = half * kNW + wave;   // the 32-column block of this wave
  const unsigned w0off = (unsigned)colblk0 * KS * 1024u;

  f32x16 acc[kRB3][CB];
#pragma unroll
  for (int rb = 0; rb < kRB3; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;

  UnitH u0;   // table rows of row brow for the chunk that is built next
  auto uload = [&](UnitH& u, const unsigned vd, const unsigned vs, const int cq) {
    const int c = cq < NC ? cq : NC - 1;                    // past the end: a harmless repeat of the last chunk
    const unsigned kb = (unsigned)c * kKC3 * 2u;
    unith_load(u, rs_tab, vd, vs, offP + kb, offQ + kb);
  };
  // chunks 0 and 1 are built up front (ring slots 0 and 1)
  uload(u0, vdst0, vsrc0, 0);
  unith_finish(u0, s_wd + kg * 8, d2r0, slot0);
  uload(u0, vdst0, vsrc0, 1);
  unith_finish(u0, s_wd + kKC3 + kg * 8, d2r0, slot0 + kA1_3);
  uload(u0, vdst0, vsrc0, 2);
  // weight fragments, requested BQD k-steps ahead of their use (a whole chunk for the coordinate kernel; the message
  // kernel, compiled for <= 128 VGPRs so that two workgroups share a CU, keeps 2 -- its other workgroup covers the rest)
  constexpr int BQD = 2;
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
    // The 8 activations a thread owes to chunk c+2 are finished 2 per k-step, as a software pipeline over the k-step's
    // 4 MFMA gaps (two stages per gap): independent instructions only, operands produced at least one stage earlier.
    //   0: 2 x fma_mix   1: exp2 a0   2: exp2 a1   3: 2 x add 1   4: rcp a0   5: rcp a1   6: 2 x mul   7: bf16 pack;
    //   the last k-step completes the 16-byte LDS slot and re-requests the consumed table row.
    float pu[2], pe[2];
#define STAGE(S, Q)                                                                                           \
    if (build) {                                                                                              \
      constexpr int e0_ = 2 * (S);   /* first element of this k-step inside the unit */                       \
      if ((Q) == 0) {                                                                                         \
        if ((S) == 0) t = u0.p + u0.q;                                                                        \
        pu[0] = fmaf(wdc[e0_], d2r0, (float)t[e0_]);                                                          \
        pu[1] = fmaf(wdc[e0_ + 1], d2r0, (float)t[e0_ + 1]);                                                  \
      }                                                                                                       \
      if ((Q) == 1) pe[0] = __builtin_amdgcn_exp2f(pu[0]);                                                    \
      if ((Q) == 2) pe[1] = __builtin_amdgcn_exp2f(pu[1]);                                                    \
      if ((Q) == 3) { pe[0] = 1.0f + pe[0]; pe[1] = 1.0f + pe[1]; }                                           \
      if ((Q) == 4) pe[0] = __builtin_amdgcn_rcpf(pe[0]);                                                     \
      if ((Q) == 5) pe[1] = __builtin_amdgcn_rcpf(pe[1]);                                                     \
      if ((Q) == 6) { pu[0] = pu[0] * pe[0]; pu[1] = pu[1] * pe[1]; }                                         \
      if ((Q) == 7) {                                                                                         \
        o[e0_] = (__bf16)pu[0];                                                                               \
        o[e0_ + 1] = (__bf16)pu[1];                                                                           \
        if ((S) == 3) { *reinterpret_cast<bf16x8*>(slot0 + off_wr) = o; uload(u0, vdst0, vsrc0, c + 3); }     \
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

  {
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
      for (int w = 0; w < kNW; ++w) v += s_part[w * kR3 + tid];
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
  }
  STAMP(31, 0);   // epilogue done
}

}  // namespace

int init_edge_bf16_x16_attributes() {
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_x16), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
  return EGNN_OK;
}

bool edge_bf16_x16_supported(const EdgeParams& p) {
  return (p.WxP == 512 || p.WxP == 1024) && x16_smem_bytes(p.WxP, p.MP, false) <= 160 * 1024 &&
         (size_t)p.N * p.TC * 4 < ((size_t)1 << 32);
}

// coordinate branch only: tiles x (WxP / 512) workgroups of 1024 threads
int launch_edge_bf16_x16(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  hipLaunchKernelGGL(edge_kernel_bf16_x16, dim3(tiles * (p.WxP / 512)), dim3(kT3), x16_smem_bytes(p.WxP, p.MP, false), st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace egnn
