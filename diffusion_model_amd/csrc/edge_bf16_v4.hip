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

#include "diag.h"
#include "edge_tile.h"

namespace egnn {

namespace {

#ifdef EGNN_EXP_WGSTAMP   // diagnostic build (diag.h, tools/fwd_stamps.py)
__device__ unsigned long long g_mwg_stamps[20000][6];
#define WG_STAMP(k) DIAG_WG_STAMP(g_mwg_stamps, 20000, k)
#define WG_STAMP_HW() DIAG_WG_STAMP_HW(g_mwg_stamps, 20000, 5)
#else
#define WG_STAMP(k)
#define WG_STAMP_HW()
#endif

constexpr int kT3 = 512;
constexpr int kR3 = 128, kRB3 = 4, kRPAD3 = kR3 + 1;
constexpr int kKC3 = 64;
constexpr size_t kA1_3 = (size_t)8 * kRPAD3 * 16;  // one activation chunk [8 k-groups][129][8 bf16]
using namespace tile128;
constexpr size_t kOffA1 = kOffLoop;      // ring of 3 activation chunks, then wd[KP] (the per-tile arrays: edge_tile.h)
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
// V8 = the MFMA operand type: bf16x8 (precision bf16) or f16x8 (precision fp16, plain forward only; kernels.h).
template <int CB, bool IS_M, bool BWD = false, bool SAVE = false, typename V8 = bf16x8>
__global__ __launch_bounds__(kT3, ((CB == 1 && !BWD) ? EGNN_V4_M_WAVES : 2)) void edge_kernel_bf16_v4(const EdgeParams p) {
  static_assert(!((BWD || SAVE) && OpTraits<V8>::f16), "the training kernels keep bf16 activations");
  typedef typename OpTraits<V8>::elem elem;
  if constexpr (OpTraits<V8>::f16) f16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lds L(smem);
  int* const s_dst = L.dst; int* const s_src = L.src;
  float* const s_d2 = L.d2; float* const s_val = L.val; float* const s_part = L.part; float* const s_gseg = L.gseg;
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

  DIAG_STAMP_SETUP(p.stamps + ((size_t)(IS_M ? 1 : 0) * 8 + wave) * 32 * 4);
  DIAG_STAMP(30, 0);   // kernel entry
  WG_STAMP(0);
  prologue_rows(p, L, e0, nvalid, IS_M ? p.wdm : p.wdx, KP, s_wd, tid);

  DIAG_STAMP(30, 1);   // tile structure ready
  WG_STAMP(1);

  // ---- K-loop ----
  const int NC = KP / kKC3, KS = KP / 16;
  const int brow = tid >> 3, kg = tid & 7;   // this thread builds rows brow and brow + 64, columns [8 kg, 8 kg + 8) of a chunk
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
  auto s1_store = [&](const V8 ov, const int row, const int c) {
    if (half != 0) return;
    if (row < nvalid)
      *reinterpret_cast<V8*>(static_cast<__bf16*>(p.s1_out) + (size_t)(e0 + row) * KP + c * kKC3 + kg * 8) = ov;
  };
  // chunks 0 and 1: all four table pieces and the first weight fragments are requested, THEN the segment structure of the tile
  // is worked out (two barriers, waves 0 and 1 only) while they are in flight, then the activations are finished (as in
  // edge_x_m16.hip; the segment modes' row_ptr loads ride under the first chunk of the K loop)
  UnitH u2, u3;
  uload(u0, vdst0, vsrc0, 0); uload(u1, vdst1, vsrc1, 0);
  uload(u2, vdst0, vsrc0, 1); uload(u3, vdst1, vsrc1, 1);
  // weight fragments, requested BQD k-steps ahead of their use (a whole chunk for the coordinate kernel; the message
  // kernel, compiled for <= 128 VGPRs so that two workgroups share a CU, keeps 2 -- its other workgroup covers the rest)
  constexpr int BQD = (CB == 1 && !BWD && EGNN_V4_M_WAVES >= 4) ? 2 : 4;
  V8 bq[BQD][CB];
#pragma unroll
  for (int s = 0; s < BQD; ++s)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) bq[s][cb] = ldbuf_v8<V8>(rs_w, lane16, w0off + ((unsigned)cb * KS + s) * 1024u);
  const int S = prologue_segments<false>(p, L, e0, nvalid, tid, lane, wave);
  {
    const V8 o0 = unith_finish<V8>(u0, s_wd + kg * 8, d2r0, slot0);
    const V8 o1 = unith_finish<V8>(u1, s_wd + kg * 8, d2r1, slot1);
    if constexpr (BWD || SAVE) { s1_store(o0, brow, 0); s1_store(o1, brow + 64, 0); }
  }
  {
    const V8 o0 = unith_finish<V8>(u2, s_wd + kKC3 + kg * 8, d2r0, slot0 + kA1_3);
    const V8 o1 = unith_finish<V8>(u3, s_wd + kKC3 + kg * 8, d2r1, slot1 + kA1_3);
    if constexpr (BWD || SAVE) { s1_store(o0, brow, 1); s1_store(o1, brow + 64, 1); }
  }
  uload(u0, vdst0, vsrc0, 2); uload(u1, vdst1, vsrc1, 2);
  const int my_mode = tid < S ? segment_mode(p, L, e0, tid) : 0;   // stored after the first chunk
  __syncthreads();

  // ring offsets (bytes): chunk c is read at off_cur, chunk c+1 at off_nxt, chunk c+2 is written at off_wr
  unsigned off_cur = 0u, off_nxt = (unsigned)kA1_3, off_wr = 2u * (unsigned)kA1_3;
  V8 a[kRB3];
#define LDS_RD(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define LDS_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
  LDS_RD(a[0], lds_a1_base, 0); LDS_RD(a[1], lds_a1_base, 512); LDS_RD(a[2], lds_a1_base, 1024); LDS_RD(a[3], lds_a1_base, 1536);

  // One chunk.  16 groups (k-step s, row block rb): wait for a[rb] (LDS returns in order: the 3 younger refills may
  // still fly), CB MFMAs, refill a[rb] in place for the next k-step (from the NEXT chunk's buffer after the last
  // k-step), and ONE activation of chunk c+2 in two halves of ~16 issue cycles, each behind an MFMA: an in-order wave can
  // only use the 24 free issue cycles of an MFMA's 32 if the fillers sit BETWEEN two MFMAs in program order.
  // sched_barrier(0) pins this order; inside a half the compiler schedules freely.
  auto chunk = [&](const int c, const bool build, const bool last) {
    DIAG_STAMP(c, 0);
    const unsigned abase = lds_a1_base + off_cur, nbase = lds_a1_base + off_nxt;
    f16x8 t;
    V8 o;
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
        _Pragma("unroll") for (int k = 0; k < 4; ++k) o[e0_ + k] = (elem)pu[k];                             \
        if ((S) == 1) { *reinterpret_cast<V8*>(slot0 + off_wr) = o; uload(u0, vdst0, vsrc0, c + 3);       \
                        if constexpr (BWD || SAVE) s1_store(o, brow, c + 2); }                                        \
        if ((S) == 3) { *reinterpret_cast<V8*>(slot1 + off_wr) = o; uload(u1, vdst1, vsrc1, c + 3);       \
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
      acc[RB][0] = mfma32(a[RB], bq[(S) % BQD][0], acc[RB][0]);                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      STAGE(S, 2 * (RB))                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      _Pragma("unroll") for (int cb = 1; cb < CB; ++cb)                                                       \
        acc[RB][cb] = mfma32(a[RB], bq[(S) % BQD][cb], acc[RB][cb]);                                          \
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
        bq[(S) % BQD][cb] = ldbuf_v8<V8>(rs_w, lane16, w0off + (unsigned)cb * KS * 1024u + ksn);              \
    }
    KSTEP(0) KSTEP(1)
    DIAG_STAMP(c, 1);
    KSTEP(2) KSTEP(3)
    DIAG_STAMP(c, 2);
#undef KSTEP
#undef GROUP
#undef STAGE
    const unsigned tmp = off_cur; off_cur = off_nxt; off_nxt = off_wr; off_wr = tmp;
  };
  // (the steady-state loop body is branch-free so that hipcc's waitcnt insertion keeps counted vmcnt waits across
  // the back edge instead of draining the queue at every control-flow join)
  DIAG_STAMP(30, 2);   // chunks 0 and 1 built, first weights requested
  WG_STAMP(2);
  DIAG_RSTAMP(31, 1);
  for (int c = 0; c < NC - 2; ++c) {
    chunk(c, true, false);
    if (c == 0 && tid < S) L.seg_mode[tid] = my_mode;
    __syncthreads(); DIAG_STAMP(c, 3);
  }
  chunk(NC - 2, false, false);
  __syncthreads();
  DIAG_STAMP(NC - 2, 3);
  chunk(NC - 1, false, true);
  __syncthreads();
  DIAG_STAMP(NC - 1, 3);
  DIAG_RSTAMP(31, 2);
#undef LDS_WAIT
#undef LDS_RD
  DIAG_STAMP(30, 3);   // K loop done
  WG_STAMP(3);


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
    static_assert(IS_M, "edge_bf16_v4.hip keeps the message kernels only (coordinate kernels: edge_x_m16.hip, edge_bf16_v3.hip)");
  } else {
    // ---- mlp_m epilogue: m = SiLU(acc + b2), gate = sigmoid(wa . m + ba) (:31-34, :59-60) ----
    static_assert(!IS_M || CB == 1, "message epilogue assumes one 32-column block per wave");
    if constexpr (SAVE) {   // scaled pre-activations to HBM first (the ring is free behind the last chunk's barrier)
      const int ncol = 32 * wave + r;
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
    message_epilogue<SAVE>(p, L, acc, S, tile, tid, lane, wave, kNegLog2e / OpTraits<V8>::wscale);
  }
  DIAG_STAMP(31, 0);   // epilogue done
  WG_STAMP(4);
  WG_STAMP_HW();
}

template <int CB, bool IS_M, bool BWD = false, bool SAVE = false, typename V8 = bf16x8>
int launch_v4(const EdgeParams& p, int blocks, size_t smem, hipStream_t st) {
  hipLaunchKernelGGL((edge_kernel_bf16_v4<CB, IS_M, BWD, SAVE, V8>), dim3(blocks), dim3(kT3), smem, st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace

int init_edge_bf16_v4_attributes() {
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v4<1, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v4<1, true, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v4<1, true, false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v4<1, true, false, false, f16x8>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return EGNN_OK;
}

int edge_v4_rows() { return kR3; }

bool edge_bf16_v4_supported(const EdgeParams& p) {
  return (p.WxP == 256 || p.WxP == 512 || p.WxP == 1024) && p.MP == 256 && p.WmP % 64 == 0 && p.WmP >= 192 &&   // >= 3 chunks: the K loop has a steady-state iteration (segment modes are stored there)
         v4_smem_bytes(p.WmP, p.MP, true) <= 160 * 1024 && v4_smem_bytes(p.WxP, p.MP, false) <= 160 * 1024 &&
         (size_t)p.N * p.TC * 4 < ((size_t)1 << 32);
}

// backward recompute of the message branch over the chunk of edges described by p
int launch_edge_bf16_v4_m_bwd(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  static_assert(8 * 32 * 72 * 2 <= kRing * kA1_3, "store staging must fit the K-loop buffers");
  return launch_v4<1, true, true>(p, tiles, v4_smem_bytes(p.WmP, p.MP, true), st);
}

// training forward of the message branch: p.s1_out / p.g_a2_out receive the activations and the scaled pre-activations
int launch_edge_bf16_v4_m_save(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  return launch_v4<1, true, false, true>(p, tiles, v4_smem_bytes(p.WmP, p.MP, true), st);
}

// message kernel, precision fp16: p.w2m = the fp16 fragment stream (scaled by -2^8 / log2(e))
int launch_edge_f16_v4_m(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  return launch_v4<1, true, false, false, f16x8>(p, tiles, v4_smem_bytes(p.WmP, p.MP, true), st);
}

// message kernel only
int launch_edge_bf16_v4_m(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR3 - 1) / kR3;
  return launch_v4<1, true>(p, tiles, v4_smem_bytes(p.WmP, p.MP, true), st);
}

}  // namespace egnn

#ifdef EGNN_EXP_WGSTAMP
extern "C" int egnn_debug_mwg_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(egnn::g_mwg_stamps), sizeof(unsigned long long) * 20000 * 6) == hipSuccess ? 0 : -1;
}
#endif
