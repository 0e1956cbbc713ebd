// Diagnostics of the hand-written kernels, in ONE place.  The product build (make) defines none of this: every macro below
// expands to nothing and every diag:: flag is false, so the kernels compile as if the lines were not there.  A diagnostic
// build (tools/exp_build.sh <name> -DEGNN_DIAG -DEGNN_EXP_...) is a separate library selected with EGNN_LIB=...; its
// results are wrong by construction for the timing switches and right for the stamp builds.
//
//   stamps      EGNN_EXP_STAMP / _STAMP2   s_memtime / s_memrealtime stamps of ONE workgroup's waves into EdgeParams::stamps
//               EGNN_EXP_WGSTAMP           100 MHz wall stamps of EVERY workgroup's phases + the CU it ran on (forward edge kernels)
//               EGNN_EXP_DGSTAMP           the same for the dgrad kernel
//   timing      EGNN_EXP_NO_S1 / _NO_T2   (training forward: drop the activation / pre-activation stores; the round-2 switches
//               _NO_BUILD / _NO_EPI / _NO_MFMA / _W_ONCE of the 32x32x16 kernels were removed with round 4: results in
//               profiles/r02i_x_decomposition.txt, code in the history before commit "diag.h")
//               EGNN_EXP_DG_NOK / _NOG / _NOW / _L2G / _NOTAB / _NOEPI               (dgrad kernel)
//               EGNN_EXP_DGG_NOSILU / _NOHOT / _NOROW (+ _DG_NOK / _DG_NOEPI)        (per-graph dgrad kernel)
//               EGNN_EXP_C8_NOCORR / _NOMAIN / _NOBUILD / _NOCVT8 / _NOEPI           (f16c8 edge kernels, + EGNN_DEBUG bits 0 / 1)
#pragma once

#if !defined(EGNN_DIAG) && (defined(EGNN_EXP_STAMP) || defined(EGNN_EXP_STAMP2) || defined(EGNN_EXP_WGSTAMP) || defined(EGNN_EXP_DGSTAMP) || \
    defined(EGNN_EXP_NO_S1) || defined(EGNN_EXP_NO_T2) || defined(EGNN_EXP_NO_STAGE) || defined(EGNN_EXP_DG_NOK) || defined(EGNN_EXP_DG_NOG) || defined(EGNN_EXP_DG_NOW) || \
    defined(EGNN_EXP_DG_L2G) || defined(EGNN_EXP_DG_NOTAB) || defined(EGNN_EXP_DG_NOEPI) || defined(EGNN_EXP_NP_NONORM) ||                    \
    defined(EGNN_EXP_NP_NOMLP) || defined(EGNN_EXP_DGG_NOSILU) || defined(EGNN_EXP_DGG_NOHOT) || defined(EGNN_EXP_DGG_NOROW) ||                   \
    defined(EGNN_EXP_C8_NOCORR) || defined(EGNN_EXP_C8_NOMAIN) || defined(EGNN_EXP_C8_NOBUILD) || defined(EGNN_EXP_C8_NOCVT8) || defined(EGNN_EXP_C8_NOEPI))
#error "EGNN_EXP_* switches are diagnostic builds: add -DEGNN_DIAG (tools/exp_build.sh)"
#endif

namespace egnn {
namespace diag {
#define EGNN_DIAG_FLAG(name, macro) constexpr bool name = macro
#ifdef EGNN_EXP_NO_S1
EGNN_DIAG_FLAG(kNoS1, true);
#else
EGNN_DIAG_FLAG(kNoS1, false);
#endif
#ifdef EGNN_EXP_NO_T2
EGNN_DIAG_FLAG(kNoT2, true);
#else
EGNN_DIAG_FLAG(kNoT2, false);
#endif
#ifdef EGNN_EXP_NO_STAGE   // training forward: skip the whole pre-activation staging block (scale pass, LDS transpose, stores)
EGNN_DIAG_FLAG(kNoStage, true);
#else
EGNN_DIAG_FLAG(kNoStage, false);
#endif
#ifdef EGNN_EXP_DG_NOK
EGNN_DIAG_FLAG(kDgNoK, true);
#else
EGNN_DIAG_FLAG(kDgNoK, false);
#endif
#ifdef EGNN_EXP_DG_NOG
EGNN_DIAG_FLAG(kDgNoG, true);
#else
EGNN_DIAG_FLAG(kDgNoG, false);
#endif
#ifdef EGNN_EXP_DG_NOW
EGNN_DIAG_FLAG(kDgNoW, true);
#else
EGNN_DIAG_FLAG(kDgNoW, false);
#endif
#ifdef EGNN_EXP_DG_L2G
EGNN_DIAG_FLAG(kDgL2G, true);
#else
EGNN_DIAG_FLAG(kDgL2G, false);
#endif
#ifdef EGNN_EXP_DG_NOTAB
EGNN_DIAG_FLAG(kDgNoTab, true);
#else
EGNN_DIAG_FLAG(kDgNoTab, false);
#endif
#ifdef EGNN_EXP_DG_NOEPI
EGNN_DIAG_FLAG(kDgNoEpi, true);
#else
EGNN_DIAG_FLAG(kDgNoEpi, false);
#endif
// per-graph dgrad (edge_bwd_dgrad_graph.hip; it also honours _DG_NOK and _DG_NOEPI): without SiLU', without the one-hot products,
// without the row sums
#ifdef EGNN_EXP_DGG_NOSILU
EGNN_DIAG_FLAG(kDggNoSilu, true);
#else
EGNN_DIAG_FLAG(kDggNoSilu, false);
#endif
#ifdef EGNN_EXP_DGG_NOHOT
EGNN_DIAG_FLAG(kDggNoHot, true);
#else
EGNN_DIAG_FLAG(kDggNoHot, false);
#endif
#ifdef EGNN_EXP_DGG_NOROW
EGNN_DIAG_FLAG(kDggNoRow, true);
#else
EGNN_DIAG_FLAG(kDggNoRow, false);
#endif
#ifdef EGNN_EXP_NP_NONORM   // node_post timing builds (tools/lat_ab.sh): without the normaliser / coordinate update, without the MLP
EGNN_DIAG_FLAG(kNpNoNorm, true);
#else
EGNN_DIAG_FLAG(kNpNoNorm, false);
#endif
#ifdef EGNN_EXP_NP_NOMLP
EGNN_DIAG_FLAG(kNpNoMlp, true);
#else
EGNN_DIAG_FLAG(kNpNoMlp, false);
#endif
// f16c8 edge kernels (edge_f16c8.hip, tools/c8_ab.sh): without the correction MFMAs (and their operand reads), without the fp16
// MFMAs, without the activation build's arithmetic (the LDS images keep their first contents), without the e4m3 conversions and
// their LDS stores, without the epilogue
#ifdef EGNN_EXP_C8_NOCORR
EGNN_DIAG_FLAG(kC8NoCorr, true);
#else
EGNN_DIAG_FLAG(kC8NoCorr, false);
#endif
#ifdef EGNN_EXP_C8_NOMAIN
EGNN_DIAG_FLAG(kC8NoMain, true);
#else
EGNN_DIAG_FLAG(kC8NoMain, false);
#endif
#ifdef EGNN_EXP_C8_NOBUILD
EGNN_DIAG_FLAG(kC8NoBuild, true);
#else
EGNN_DIAG_FLAG(kC8NoBuild, false);
#endif
#ifdef EGNN_EXP_C8_NOCVT8
EGNN_DIAG_FLAG(kC8NoCvt8, true);
#else
EGNN_DIAG_FLAG(kC8NoCvt8, false);
#endif
#ifdef EGNN_EXP_C8_NOEPI
EGNN_DIAG_FLAG(kC8NoEpi, true);
#else
EGNN_DIAG_FLAG(kC8NoEpi, false);
#endif
#undef EGNN_DIAG_FLAG
// EdgeParams::dbg (environment EGNN_DEBUG, read by a diagnostic build only): bit 0 = zero-size weight descriptor, bit 1 =
// zero-size table descriptor (loads that never leave the CU), bit 2 = every chunk reads chunk 0's weight fragments (edge_f16c8w.hip:
// a weight stream that always hits the L2).  The product build ignores the field at compile time.
#ifdef EGNN_DIAG
__host__ __device__ inline bool drop_weight_loads(int dbg) { return (dbg & 1) != 0; }
__host__ __device__ inline bool drop_table_loads(int dbg) { return (dbg & 2) != 0; }
__host__ __device__ inline bool hot_weight_loads(int dbg) { return (dbg & 4) != 0; }
#else
__host__ __device__ constexpr bool hot_weight_loads(int) { return false; }
__host__ __device__ constexpr bool drop_weight_loads(int) { return false; }
__host__ __device__ constexpr bool drop_table_loads(int) { return false; }
#endif
}  // namespace diag
}  // namespace egnn

// ---- per-wave stamps of one workgroup (tools/stamps.py layout: [kernel][wave][32 slots][4]) ----------------------------
// DIAG_STAMP_SETUP(base) once per kernel (base = this wave's block of EdgeParams::stamps; needs `lane` in scope), then
// DIAG_STAMP(slot, k) = s_memtime, DIAG_RSTAMP(slot, k) = s_memrealtime (100 MHz: clock = d cycles / d wall x 100 MHz).
#ifdef EGNN_EXP_STAMP
#define DIAG_STAMP_SETUP(base)                        \
  const bool diag_stamp_wg = blockIdx.x == gridDim.x / 2; \
  unsigned long long* const diag_st_base = (base)
#define DIAG_STAMP_(insn, c, k)                                                                   \
  do {                                                                                            \
    unsigned long long t_;                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    asm volatile(insn " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (diag_stamp_wg && lane == 0 && (c) < 32) diag_st_base[(c) * 4 + (k)] = t_;                 \
  } while (0)
#define DIAG_STAMP(c, k) DIAG_STAMP_("s_memtime", c, k)
#define DIAG_RSTAMP(c, k) DIAG_STAMP_("s_memrealtime", c, k)
#else
#define DIAG_STAMP_SETUP(base)
#define DIAG_STAMP(c, k)
#define DIAG_RSTAMP(c, k)
#endif
#ifdef EGNN_EXP_STAMP2   // finer stamps inside the first half of a chunk (replaces the meaning of slots 1..3)
#define DIAG_STAMP2(c, k, cond) do { if (cond) DIAG_STAMP(c, k); } while (0)
#define DIAG_STAMP1(c, k)
#else
#define DIAG_STAMP2(c, k, cond)
#define DIAG_STAMP1(c, k) DIAG_STAMP(c, k)
#endif

// ---- every workgroup's phase stamps into a __device__ array `arr[blocks][slots]` (tools/fwd_stamps.py, dgrad_stamps.py) ---
#if defined(EGNN_EXP_WGSTAMP) || defined(EGNN_EXP_DGSTAMP)
#define DIAG_WG_STAMP(arr, nblocks, k)                                                                \
  do {                                                                                                \
    unsigned long long t_;                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (threadIdx.x == 0 && blockIdx.x < (nblocks)) (arr)[blockIdx.x][k] = t_;                        \
  } while (0)
#define DIAG_WG_STAMP_HW(arr, nblocks, k)                                                             \
  do {                                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < (nblocks))                                                   \
      (arr)[blockIdx.x][k] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |          \
                             ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);   \
  } while (0)
#endif
