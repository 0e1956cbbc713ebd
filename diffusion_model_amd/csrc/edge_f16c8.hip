// Per-edge kernels of precision 'f16c8' (gfx950): fp32-grade accuracy for TWO bf16-equivalents of matrix work per product.
//
// north_star asks for eps within 1e-4 of the reference's fp32 path (EquivariantGraphNeuralNetwork.py:55-65 is fp32
// throughout).  Precision 'bf16x3' meets it with three full-rate products per operand pair (edge_bf16x3.hip).  Here both
// operands of the two per-edge second-layer products (:15-16 mlp_m.2, :21-22 mlp_x.2) are split into an fp16 head (11
// significant bits) and a remainder,          a = a_hi + a_lo,      W = W_hi + W_lo,
//     a . W  ~=  a_hi . W_hi                          v_mfma_f32_16x16x32_f16      (full rate, exact products, fp32 accumulate)
//              + a_lo . W_hi  +  a_hi . W_lo          v_mfma_scale_f32_16x16x128_f8f6f4 on e4m3 operands (TWICE the f16 rate)
// and the two remainder products run on OCP e4m3 (4 significant bits): a remainder is 2^-12 of its operand, so the e4m3
// rounding of the correction's operands is 2^-16 of the product -- what bf16x3's dropped term and remainder rounding
// cost (2^-17).  Both corrections share ONE block-scaled instruction per 16 x 16 tile and 64 hidden units: its K = 128 is
// four blocks of 32 with a scale each, laid out as [a_lo8 | a_hi8 | a_lo8 | a_hi8] against [W_hi8 | W_lo8 | W_hi8 | W_lo8]
// (hidden units 0-31, 0-31, 32-63, 32-63 of the chunk), and the e8m0 block scales carry the FIXED powers of two the
// operands were multiplied by before rounding (2^12 for a_lo, 2 for a_hi8; per-matrix exponents for the weights, c8_exp),
// so the instruction adds straight into the main product's accumulators.  No dynamic (per-block) scale is needed: e4m3
// spans 2^-9 ... 448, and whatever underflows is 2^-21 of the largest activation the scale admits.
// Measured (tools/micro/mfma_scale_probe.hip, profiles/r05a_mfma_scale_probe.txt): the mix {2 f16 + 1 scaled e4m3} takes
// 64 SIMD-cycles per tile and chunk against 96.5 for bf16x3's six bf16 MFMAs, at a higher held clock (1.96 vs 1.85 GHz).
// CPU emulation of every rounding (tools/rounding_budget.py, profiles/r05_rounding_budget.txt): 6e-6 / 1e-5 on h' / eps_x
// of the full-width goldens (bar 1e-4).
//
// Everything else is the fp32 path's arithmetic: fp32 first-layer table (exact f32 MFMA in node_pre), fp32 geometry, SiLU /
// sigmoid, heads and segment sums; split-operand node MLP (node_bf16.hip).  Tile and K loop are those of edge_x_m16.hip
// (128 edges, 8 waves, 16x16 MFMA tiles, 64-deep chunks, phase-opposed SIMD partners, XOR-swizzled fp16 image) with
//   IS_M = false  coordinate branch (:62-65): 512 columns of mlp_x.2 per workgroup (WxP / 512 column shares), 4 column blocks per wave
//   IS_M = true   message branch (:57-61): all 256 columns of mlp_m.2 + the attention gate, 2 column blocks per wave
// Request distances as in that kernel: the weight fragments (fp16 and e4m3) of chunk c + 1 are requested at the start of the
// matrix phase of chunk c (two register sets), the fp32 table rows of a build right after the previous build.  That fits beside
// the accumulators at 2 column blocks per wave (256 columns per workgroup, 64 accumulator registers); at 4 column blocks (512
// columns: the mlp_x activations built twice per tile instead of four times) the operands of the second matrix instruction do
// not fit beside 128 accumulators with those distances -- hipcc spills inside the K loop -- so the coordinate branch (kCBX = 4)
// runs the "tight" schedule of edge_c8_kernel: every operand group requested as late as its latency allows.
// Since r05C the default f16c8 kernels are the 32x32-tile ones of edge_f16c8w.hip; these run with EGNN_C8_TILE=16 and for message
// widths that are a multiple of 64 but not of 128.
#include "diag.h"
#include "edge_tile.h"
#include <type_traits>

namespace egnn {

namespace {

using namespace tile128;
constexpr int kT = 512;
constexpr int kKC = 64;                        // activation chunk depth
#ifndef C8_CBX      // (A/B builds: tools/c8_ab2.sh)
#define C8_CBX 4
#endif
#ifndef C8_VPRIO    // s_setprio of a wave in its build / in its matrix phase
#define C8_VPRIO 3
#endif
#ifndef C8_MPRIO
#define C8_MPRIO 0
#endif
#ifndef C8_MPHASE4_INC   // generated body of the 4-column-block matrix phase and its ring depths (tools/gen/gen_c8_mphase.py)
#define C8_MPHASE4_INC "edge_f16c8_mphase4.inc"
#define C8_RA4 3
#define C8_RC4 2
#endif
constexpr int kCBX = C8_CBX;                        // coordinate kernel: 16-column blocks per wave (4: 512 columns per workgroup; see edge_c8_kernel)
constexpr size_t kA1 = (size_t)8 * kR * 16;    // fp16 image of a chunk: [8 k-groups][128 rows][8 f16], rows XOR-swizzled (edge_x_m16.hip)
// e4m3 image of a chunk: four K blocks [a_lo8 0-31 | a_hi8 0-31 | a_lo8 32-63 | a_hi8 32-63] of [128 rows][32 B], block q at
// q * 4096 + 16 (q & 1) + 64 (q >> 1).  No swizzle: an operand read (ds_read_b128, 16-lane groups = rows 0-3 and 12-15 of block q
// + rows 4-11 of block q + 1) finds the even block's rows on the even and the odd block's on the odd 16-byte slots of the 256-B
// load bank row; a store instruction (ds_write_b64, 16 lanes = 2 rows x 8 pieces, blocks q and q + 2) covers the two 64-B
// halves of the 128-B store bank row.  (The first build shifted whole blocks by 32 B and swapped halves by block parity:
// SQ_LDS_BANK_CONFLICT = 26 % of the LDS cycles, profiles/r05c_f16c8_profile_summary.txt.)
__host__ __device__ constexpr size_t c8_block(int q) { return (size_t)q * 4096 + 16 * (q & 1) + 64 * (q >> 1); }
constexpr size_t kC8 = 4 * 4096 + 128;
__host__ __device__ inline size_t c8_smem_bytes(int KP, bool is_m) {
  (void)is_m;
  return kOffLoop + 2 * kA1 + 2 * kC8 + (size_t)KP * 4;
}

typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) short i16x2;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// SiLU + the three operand forms of one build unit (8 hidden units of one row): table, wd pre-scaled by -log2(e).
//   slot16: 16 B of the fp16 image; slot_lo / slot_hi: 8 B each of the e4m3 image (remainder x 2^12, value x 2)
__device__ __forceinline__ void unit_finish_c8(const Unit& u, const float* wd, float d2, char* slot16, char* slot_lo, char* slot_hi) {
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(wd), w1 = *reinterpret_cast<const f32x4*>(wd + 4);
  float a[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a[j] = silu_s(fmaf(w0[j], d2, u.p0[j] + u.q0[j]));
    a[j + 4] = silu_s(fmaf(w1[j], d2, u.p1[j] + u.q1[j]));
  }
  const f16x8 h = pack8<f16x8>(a);   // RNE; MODE.FP16_OVFL: saturates instead of inf
  if constexpr (diag::kC8NoCvt8) { *reinterpret_cast<f16x8*>(slot16) = h; return; }
  float lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) lo[j] = a[j] - (float)h[j];
  // v_cvt_scalef32_pk_fp8_f32: e4m3(src / scale), RNE, saturating at +-448 under MODE.FP16_OVFL (tools/micro/mfma_scale_probe.hip)
  i16x2 l0 = {0, 0}, l1 = {0, 0}, h0 = {0, 0}, h1 = {0, 0};
  l0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l0, lo[0], lo[1], 0x1p-12f, false);
  l0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l0, lo[2], lo[3], 0x1p-12f, true);
  l1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l1, lo[4], lo[5], 0x1p-12f, false);
  l1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l1, lo[6], lo[7], 0x1p-12f, true);
  h0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h0, a[0], a[1], 0.5f, false);
  h0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h0, a[2], a[3], 0.5f, true);
  h1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h1, a[4], a[5], 0.5f, false);
  h1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h1, a[6], a[7], 0.5f, true);
  *reinterpret_cast<f16x8*>(slot16) = h;
  *reinterpret_cast<u32x2*>(slot_lo) = u32x2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
  *reinterpret_cast<u32x2*>(slot_hi) = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
}

// CB = 16-column blocks per wave: the workgroup's 8 waves cover 128 CB columns (message branch: CB = 2 = all 256 columns)
// table rows of one build unit: voffset = byte offset of the row (+ the unit's column piece), soffset = column offset of the chunk;
// the second 16-byte piece's + 16 rides in the SCALAR offset (unit_load of kernels.h adds it to the vector offset, and hipcc then
// keeps row + 16 in a register of its own across the K loop: four more address registers, which spilled)
__device__ __forceinline__ void unit_load_c8(Unit& u, rsrc_t tab, unsigned vdst, unsigned vsrc, unsigned sP, unsigned sQ) {
#ifndef C8_UNIT_SOFF   // default: the shared unit_load (+16 on the vector offset); -DC8_UNIT_SOFF = +16 on the scalar offset: measured 8 % SLOWER (profiles/r05n_c8_voff_ab.txt)
  unit_load(u, tab, vdst, vsrc, sP, sQ);
#else
  u.p0 = ldbuf_f32x4(tab, vdst, sP);
  u.p1 = ldbuf_f32x4(tab, vdst, sP + 16u);
  u.q0 = ldbuf_f32x4(tab, vsrc, sQ);
  u.q1 = ldbuf_f32x4(tab, vsrc, sQ + 16u);
#endif
}

template <bool IS_M, int CB>
__global__ __launch_bounds__(kT, 2) void edge_c8_kernel(const EdgeParams p) {
  static_assert(!IS_M || CB == 2, "the message branch is 256 columns wide");
  f16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lds L(smem);
  char* s_a1 = smem + kOffLoop;             // [2 buffers] fp16 image
  char* s_c8 = s_a1 + 2 * kA1;              // [2 buffers] e4m3 image
  float* s_wd = reinterpret_cast<float*>(s_c8 + 2 * kC8);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r15 = lane & 15, q4 = lane >> 4;
  const int KP = IS_M ? p.WmP : p.WxP;
  const int nsplit = IS_M ? 1 : p.WxP / (128 * CB);
  const int j = xcd_tile(blockIdx.x, gridDim.x);
  const int tile = j / nsplit, half = j - tile * nsplit;
  const int e0 = tile * kR;
  const int nvalid = min(kR, p.E - e0);

  DIAG_STAMP_SETUP(p.stamps + ((size_t)(IS_M ? 1 : 0) * 8 + wave) * 32 * 4);   // tools/stamps.py (diagnostic build only)
  DIAG_STAMP(30, 0);   // kernel entry

  // ---- weight streams ----
  const int NC = KP / kKC, KS = KP / 32;
  const int brow = tid >> 3, kg = tid & 7;   // this thread builds rows brow and brow + 64, hidden units 8 kg .. 8 kg + 7 of a chunk
  const size_t ncols = IS_M ? (size_t)p.MP : (size_t)p.WxP;
  const unsigned wbytes = diag::drop_weight_loads(p.dbg) ? 0u : (unsigned)(ncols * KP * 2);
  const rsrc_t rs_w = make_rsrc(IS_M ? p.w2m16 : p.w2x16, wbytes);    // fp16 fragments [N/16][K/32][64][8]
  const rsrc_t rs_w8 = make_rsrc(IS_M ? p.w2m_c8 : p.w2x_c8, wbytes);   // e4m3 fragments [N/16][K/64][2][64][16 B]
  const unsigned lane16 = lane * 16u;
  const int cb0 = IS_M ? wave * CB : half * (8 * CB) + wave * CB;   // first 16-column block of this wave
  const unsigned w0 = (unsigned)cb0 * KS * 1024u;              // fp16 stream: 1 KiB per (column block, k-step)
  const unsigned w80 = (unsigned)cb0 * NC * 2048u;             // e4m3 stream: 2 KiB per (column block, chunk)
  auto ld16 = [&](const int c, const int s, const int cb) {   // fp16 fragment of (chunk c, k-step s, column block cb)
    return ldbuf_v8<f16x8>(rs_w, lane16, w0 + ((unsigned)cb * KS + (unsigned)(2 * c + s)) * 1024u);
  };
  auto ld8 = [&](const int c, const int cb) {                 // e4m3 fragment (32 bytes per lane) of (chunk c, column block cb)
    const unsigned o = w80 + ((unsigned)cb * NC + (unsigned)c) * 2048u;
    const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rs_w8, lane16, __builtin_amdgcn_readfirstlane(o), 0);
    const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rs_w8, lane16, __builtin_amdgcn_readfirstlane(o + 1024u), 0);
    return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
  };
  // 2 column blocks: two register sets, set c & 1 for chunk c (a whole chunk of request distance); 4 column blocks: one set
  constexpr int NSET = CB == 2 ? 2 : 1;
  f16x8 bq[NSET][2][CB];   // [set][k-step][column block] fp16
  i32x8 b8[NSET][CB];      // [set][column block] e4m3
  prologue_rows(p, L, e0, nvalid, IS_M ? p.wdm : p.wdx, KP, s_wd, tid);
  DIAG_STAMP(30, 1);   // edge rows and geometry ready

  // ---- K loop ----
  const rsrc_t rs_tab = make_rsrc(p.table, diag::drop_table_loads(p.dbg) ? 0u : (unsigned)min((size_t)p.N * p.TC * 4, (size_t)0xFFFFFFFFu));
  const unsigned vdst0 = (unsigned)L.dst[brow] * (unsigned)p.TC * 4u + (unsigned)kg * 32u;
  const unsigned vsrc0 = (unsigned)L.src[brow] * (unsigned)p.TC * 4u + (unsigned)kg * 32u;
  const unsigned vdst1 = (unsigned)L.dst[brow + 64] * (unsigned)p.TC * 4u + (unsigned)kg * 32u;
  const unsigned vsrc1 = (unsigned)L.src[brow + 64] * (unsigned)p.TC * 4u + (unsigned)kg * 32u;
  const unsigned offP = (IS_M ? 2u * p.WxP : 0u) * 4u, offQ = (IS_M ? 2u * p.WxP + p.WmP : (unsigned)p.WxP) * 4u;   // fp32 table {Px|Qx|Pm|Qm}
  // fp16 image slots of this thread (rows brow, brow + 64), as edge_x_m16.hip
  char* slot0 = s_a1 + (size_t)kg * (kR * 16) + (size_t)(brow ^ kg) * 16;
  // e4m3 image: hidden units 8 kg .. + 7 sit in K block 2 (kg >> 2) (remainder) / + 1 (value), bytes 8 (kg & 3) .. + 7 of the row
  char* slo0 = s_c8 + (size_t)(kg >> 2) * (c8_block(2) - c8_block(0)) + (size_t)brow * 32 + (size_t)(kg & 3) * 8;
  constexpr int shi_delta = (int)(c8_block(1) - c8_block(0));   // the value slot: same place in the next block
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)s_a1;
  const unsigned abase0 = lds0 + (unsigned)q4 * (kR * 16) + (unsigned)(r15 ^ q4) * 16u;
  const unsigned abase1 = lds0 + (unsigned)(4 + q4) * (kR * 16) + (unsigned)(r15 ^ (4 + q4)) * 16u;
  const unsigned lds8 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)s_c8;
  const unsigned cbase = lds8 + (unsigned)q4 * 4096u + 16u * (unsigned)(q4 & 1) + 64u * (unsigned)(q4 >> 1) + (unsigned)r15 * 32u;   // c8_block(q4): this lane's 32 operand bytes of row block 0
  // block scales (e8m0, byte 0 of the scale operands): the hardware applies scale_a(row, block) * scale_b(column, block), and the
  // two kinds of block need the same product -- even blocks a_lo8 x W_hi8: 2^-12 x 2^-s_hi, odd blocks a_hi8 x W_lo8: 2^-1 x
  // 2^-(s_hi + 11) -- so both operands take ONE wave-uniform scale: a literal and a scalar register, no vector registers
  constexpr int kScaleA = 127 - 12;
  const int scale_b = __builtin_amdgcn_readfirstlane((IS_M ? p.c8_exp + 2 : p.c8_exp)[0]);   // 127 - s_hi (pack_frags_c8)

  f32x4v acc[8][CB];
#pragma unroll
  for (int rb = 0; rb < 8; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};

  Unit ua0, ua1;
  auto vload = [&](const int cq) {   // fp32 table rows for the activations of chunk cq (clamped: a harmless repeat at the end)
    const int c = cq < NC ? cq : NC - 1;
    const unsigned kb = (unsigned)c * kKC * 4u;
    unit_load_c8(ua0, rs_tab, vdst0, vsrc0, offP + kb, offQ + kb);
    unit_load_c8(ua1, rs_tab, vdst1, vsrc1, offP + kb, offQ + kb);
  };
  // SiLU + operand forms of one row of chunk c into LDS buffer PAR = c & 1 (a compile-time parity: the K loops below are
  // unrolled by two so that every LDS address is a per-thread base + an immediate)
  auto vrow = [&](auto par_c, auto row_c, const int c) {
    constexpr int PAR = decltype(par_c)::value, ROW = decltype(row_c)::value;
    unit_finish_c8(ROW ? ua1 : ua0, s_wd + c * kKC + kg * 8, L.d2[brow + 64 * ROW], slot0 + PAR * kA1 + ROW * 64 * 16,
                   slo0 + PAR * kC8 + ROW * 64 * 32, slo0 + PAR * kC8 + ROW * 64 * 32 + shi_delta);
  };
  const std::integral_constant<int, 0> P0;
  const std::integral_constant<int, 1> P1;
  int S = 0;   // segments of the tile

  // ---- the operand pipeline of a matrix phase, shared by both schedules (bodies generated: tools/gen/gen_c8_mphase.py) ----
  // LDS returns in order and every wait counts the reads issued AFTER the one it needs.  A scheduling barrier sits behind every
  // group of MFMAs: hipcc otherwise sinks the MFMA builtins past the volatile asm that follows -- the waits then run back to
  // back with the matrix work behind them, accumulators get renamed and the K loop spills.
#define LDS_RD(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define LDS_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define MAIN_STEP(A, KSTEP, RB)                                                                                          \
  do {                                                                                                                   \
    asm volatile("" : "+v"(A)); /* uses of the piece stay below the wait */                                              \
    if constexpr (!diag::kC8NoMain) {                                                                                    \
      _Pragma("unroll") for (int cb = 0; cb < CB; ++cb)                                                                  \
          acc[RB][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, BQ(KSTEP)[cb], acc[RB][cb], 0, 0, 0);                  \
    } else {                                                                                                             \
      asm volatile("" ::"v"(BQ(KSTEP)[0]), "v"(BQ(KSTEP)[CB - 1]));                                                      \
    }                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  } while (0)
#define CORR_STEP(C0, C1, RB)                                                                                            \
  do {                                                                                                                   \
    if constexpr (!diag::kC8NoCorr) {                                                                                    \
      asm volatile("" : "+v"(C0), "+v"(C1));                                                                             \
      const u32x4 x0_ = C0, x1_ = C1;                                                                                    \
      const i32x8 a8_ = {(int)x0_.x, (int)x0_.y, (int)x0_.z, (int)x0_.w, (int)x1_.x, (int)x1_.y, (int)x1_.z, (int)x1_.w}; \
      _Pragma("unroll") for (int cb = 0; cb < CB; ++cb)                                                                  \
          acc[RB][cb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8_, B8[cb], acc[RB][cb], 0, 0, 0, kScaleA, 0, scale_b); \
    } else {                                                                                                             \
      asm volatile("" ::"v"(B8[0]), "v"(B8[CB - 1]));                                                                    \
    }                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  } while (0)

  if constexpr (CB == 2) {
    // ================= 2 column blocks per wave (256 columns per workgroup): the registers allow a whole chunk of request
    // distance for everything, as in edge_x_m16.hip =================
    // weight fragments: two register sets, set c & 1 for chunk c; the set of chunk c + 1 is requested at the start of the matrix
    // phase of chunk c; the table rows of a build right behind the previous build.  An fp16 piece feeds only 32 cycles of MFMAs
    // here and an e4m3 operand 64, so the operand reads run 8 pieces / 4 operands (256 cycles) ahead of their use.
    auto wload = [&](auto par_c, const int cq) {
      constexpr int PAR = decltype(par_c)::value;
      const int c = cq < NC ? cq : NC - 1;   // past the end: a harmless repeat
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) bq[PAR][s][cb] = ld16(c, s, cb);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) b8[PAR][cb] = ld8(c, cb);
    };
    auto vfinish = [&](auto par_c, const int c) {
      if constexpr (diag::kC8NoBuild) { if (c > 1) { asm volatile("" :: "v"(ua0.p0), "v"(ua0.p1), "v"(ua0.q0), "v"(ua0.q1), "v"(ua1.p0), "v"(ua1.p1), "v"(ua1.q0), "v"(ua1.q1)); return; } }
      __builtin_amdgcn_s_setprio(C8_VPRIO);   // vector work wins issue arbitration over the partner wave's MFMAs
      vrow(par_c, P0, c);
      vrow(par_c, P1, c);
      __builtin_amdgcn_s_setprio(C8_MPRIO);
    };
    // chunk 0: table rows and the first weight fragments are requested, then the segment structure is worked out while they fly.
    // (Same-box A/Bs of other placements, each SLOWER: fragments at kernel entry -1.5 %, behind the row prologue + both rows'
    // table requests in front of the segment structure -0.8 %: profiles/r05p_c8_entry_ab.txt, r05z_c8_prologue_ab.txt.)
    vload(0);
    wload(P0, 0);
    S = prologue_segments<false>(p, L, e0, nvalid, tid, lane, wave);
    vfinish(P0, 0);
    vload(1);
    __syncthreads();
    DIAG_STAMP(30, 2);   // chunk 0 built, first weights requested
    DIAG_RSTAMP(31, 1);
    auto mphase = [&](auto par_c, auto npar_c, const int c, const bool last) {
      constexpr int PAR = decltype(par_c)::value;
      constexpr int kO16 = PAR * (int)kA1, kO8 = PAR * (int)kC8;
      f16x8 a[8];
      u32x4 c0[4], c1[4];
#define BQ(s) bq[PAR][s]
#define B8 b8[PAR]
#define MPHASE_AFTER_FIRST_READS if (!last) wload(npar_c, c + 1)
#define MPHASE_AFTER_KSTEP0
#define MPHASE_AFTER_KSTEP1
#include "edge_f16c8_mphase2.inc"
#undef MPHASE_AFTER_KSTEP1
#undef MPHASE_AFTER_KSTEP0
#undef MPHASE_AFTER_FIRST_READS
#undef B8
#undef BQ
    };
    // SIMD partners (waves w and w + 4) in opposite phase, one barrier per chunk (edge_x_m16.hip); two chunks per loop iteration
    // (NC is even: edge_f16c8_supported)
    if (wave < 4) {   // multiply chunk i, then build chunk i + 1 and request the table rows of chunk i + 2
      const int my_mode = tid < S ? segment_mode(p, L, e0, tid) : 0;   // row_ptr loads of the segment modes: under the first matrix phase
      DIAG_STAMP(0, 0);
      mphase(P0, P1, 0, false);
      DIAG_STAMP(0, 1);
      if (tid < S) L.seg_mode[tid] = my_mode;
      vfinish(P1, 1); vload(2);
      DIAG_STAMP(0, 2);
      __syncthreads();
      DIAG_STAMP(0, 3);
      for (int i = 1; i + 1 < NC - 1; i += 2) {
        DIAG_STAMP(i, 0); mphase(P1, P0, i, false); DIAG_STAMP(i, 1); vfinish(P0, i + 1); vload(i + 2); DIAG_STAMP(i, 2); __syncthreads(); DIAG_STAMP(i, 3);
        DIAG_STAMP(i + 1, 0); mphase(P0, P1, i + 1, false); DIAG_STAMP(i + 1, 1); vfinish(P1, i + 2); vload(i + 3); DIAG_STAMP(i + 1, 2); __syncthreads(); DIAG_STAMP(i + 1, 3);
      }
    } else {          // build chunk i + 1 first, then multiply chunk i
      DIAG_STAMP(0, 0);
      vfinish(P1, 1); vload(2);
      DIAG_STAMP(0, 1);
      __builtin_amdgcn_sched_barrier(0);
      mphase(P0, P1, 0, false);
      DIAG_STAMP(0, 2);
      __syncthreads();
      DIAG_STAMP(0, 3);
      for (int i = 1; i + 1 < NC - 1; i += 2) {
        DIAG_STAMP(i, 0); vfinish(P0, i + 1); vload(i + 2); DIAG_STAMP(i, 1); __builtin_amdgcn_sched_barrier(0); mphase(P1, P0, i, false); DIAG_STAMP(i, 2); __syncthreads(); DIAG_STAMP(i, 3);
        DIAG_STAMP(i + 1, 0); vfinish(P1, i + 2); vload(i + 3); DIAG_STAMP(i + 1, 1); __builtin_amdgcn_sched_barrier(0); mphase(P0, P1, i + 1, false); DIAG_STAMP(i + 1, 2); __syncthreads(); DIAG_STAMP(i + 1, 3);
      }
    }
    DIAG_STAMP(NC - 1, 0);
    mphase(P1, P0, NC - 1, true);
    DIAG_STAMP(NC - 1, 1);
  } else {
    // ================= 4 column blocks per wave (512 columns per workgroup: the mlp_x activations are built twice per tile
    // instead of four times) =================
    // 128 accumulator registers leave 128 for everything else -- fp16 fragments (2 k-steps x 16), e4m3 fragments (32), fp32 table
    // rows (2 x 16), operand rings (12 / 16), a row's build temporaries (~30), addresses (~14) -- and hipcc needs ~40 of slack or
    // it spills inside the K loop.  So every operand group is requested as LATE as its latency allows, and no two 32-register
    // groups are ever live together:
    //   fp16 fragments, k-step 1 of chunk c      at the start of its matrix phase (used 512 cycles later)
    //   e4m3 fragments of chunk c                behind its first fp16 k-step (used 512 cycles later)
    //   table rows of the next build: row 0      behind the second fp16 k-step (used 1,024 cycles later); row 1 at the start of
    //                                            the build (used behind row 0's arithmetic)
    //   fp16 fragments, k-step 0 of the next matrix phase: between the two rows of the build in front of it
    // and the build finishes its two rows one after the other (scheduling barriers): one row's temporaries are live at a time.
    auto wload16 = [&](const int cq, const int s) {
      const int c = cq < NC ? cq : NC - 1;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) bq[0][s][cb] = ld16(c, s, cb);
    };
    auto wload8 = [&](const int c) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) b8[0][cb] = ld8(c, cb);
    };
    auto vload0 = [&](const int cq) {
      const int c = cq < NC ? cq : NC - 1;
      unit_load_c8(ua0, rs_tab, vdst0, vsrc0, offP + (unsigned)c * kKC * 4u, offQ + (unsigned)c * kKC * 4u);
    };
    auto vload1 = [&](const int cq) {
      const int c = cq < NC ? cq : NC - 1;
      unit_load_c8(ua1, rs_tab, vdst1, vsrc1, offP + (unsigned)c * kKC * 4u, offQ + (unsigned)c * kKC * 4u);
    };
    // build of chunk c (row 0's table rows were requested by the matrix phase in front of it); mchunk: the chunk of the NEXT matrix
    // phase of this wave, whose k-step-0 fragments are requested between the rows
    auto vfinish = [&](auto par_c, const int c, const int mchunk, const bool row1_requested = false) {
      __builtin_amdgcn_s_setprio(C8_VPRIO);
      if (!row1_requested) vload1(c);
      __builtin_amdgcn_sched_barrier(0);
      if (!(diag::kC8NoBuild && c > 1)) vrow(par_c, P0, c);
      else asm volatile("" :: "v"(ua0.p0), "v"(ua0.p1), "v"(ua0.q0), "v"(ua0.q1));
      __builtin_amdgcn_sched_barrier(0);
      DIAG_STAMP2(c + 15, 3, c >= 1 && c < 15);   // row 0 done (slot of chunk c - 1's matrix phase: c + 15)
      if (mchunk >= 0) wload16(mchunk, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (!(diag::kC8NoBuild && c > 1)) vrow(par_c, P1, c);
      else asm volatile("" :: "v"(ua1.p0), "v"(ua1.p1), "v"(ua1.q0), "v"(ua1.q1));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(C8_MPRIO);
    };
    vload0(0);
    S = prologue_segments<false>(p, L, e0, nvalid, tid, lane, wave);
    vfinish(P0, 0, 0);          // chunk 0 (+ the k-step-0 fragments of chunk 0, between its rows)
    if (wave >= 4) vload0(1);   // waves 4-7 build chunk 1 first thing in the loop
    __syncthreads();
    DIAG_STAMP(30, 2);
    DIAG_RSTAMP(31, 1);
    // tab_chunk: chunk whose row-0 table rows are requested behind the second k-step (-1: none)
    auto mphase = [&](auto par_c, const int c, const int tab_chunk) {
      constexpr int PAR = decltype(par_c)::value;
      constexpr int kO16 = PAR * (int)kA1, kO8 = PAR * (int)kC8;
      f16x8 a[C8_RA4];
      u32x4 c0[C8_RC4], c1[C8_RC4];
#define BQ(s) bq[0][s]
#define B8 b8[0]
#define MPHASE_AFTER_FIRST_READS do { __builtin_amdgcn_sched_barrier(0); wload16(c, 1); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MPHASE_AFTER_KSTEP0 do { DIAG_STAMP2(c + 16, 0, c < 14); wload8(c); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MPHASE_AFTER_KSTEP1 do { DIAG_STAMP2(c + 16, 1, c < 14); if (tab_chunk >= 0) vload0(tab_chunk); __builtin_amdgcn_sched_barrier(0); } while (0)
#include C8_MPHASE4_INC
      DIAG_STAMP2(c + 16, 2, c < 14);
#undef MPHASE_AFTER_KSTEP1
#undef MPHASE_AFTER_KSTEP0
#undef MPHASE_AFTER_FIRST_READS
#undef B8
#undef BQ
    };
    if (wave < 4) {   // multiply chunk i, then build chunk i + 1
      const int my_mode = tid < S ? segment_mode(p, L, e0, tid) : 0;
      DIAG_STAMP(0, 0);
      mphase(P0, 0, 1);
      DIAG_STAMP(0, 1);
      if (tid < S) L.seg_mode[tid] = my_mode;
      vfinish(P1, 1, 1);
      DIAG_STAMP(0, 2);
      __syncthreads();
      DIAG_STAMP(0, 3);
      for (int i = 1; i + 1 < NC - 1; i += 2) {
        DIAG_STAMP(i, 0); mphase(P1, i, i + 1); DIAG_STAMP(i, 1); vfinish(P0, i + 1, i + 1); DIAG_STAMP(i, 2); __syncthreads(); DIAG_STAMP(i, 3);
        DIAG_STAMP(i + 1, 0); mphase(P0, i + 1, i + 2); DIAG_STAMP(i + 1, 1); vfinish(P1, i + 2, i + 2); DIAG_STAMP(i + 1, 2); __syncthreads(); DIAG_STAMP(i + 1, 3);
      }
    } else {          // build chunk i + 1, then multiply chunk i
      DIAG_STAMP(0, 0);
      vfinish(P1, 1, 0);
      DIAG_STAMP(0, 1);
      __builtin_amdgcn_sched_barrier(0);
      mphase(P0, 0, 2);
      DIAG_STAMP(0, 2);
      __syncthreads();
      DIAG_STAMP(0, 3);
      for (int i = 1; i + 1 < NC - 1; i += 2) {
        DIAG_STAMP(i, 0); vfinish(P0, i + 1, i); DIAG_STAMP(i, 1); __builtin_amdgcn_sched_barrier(0); mphase(P1, i, i + 2); DIAG_STAMP(i, 2); __syncthreads(); DIAG_STAMP(i, 3);
        DIAG_STAMP(i + 1, 0); vfinish(P1, i + 2, i + 1); DIAG_STAMP(i + 1, 1); __builtin_amdgcn_sched_barrier(0); mphase(P0, i + 1, i + 3); DIAG_STAMP(i + 1, 2); __syncthreads(); DIAG_STAMP(i + 1, 3);
      }
      wload16(NC - 1, 0);   // the last matrix phase has no build in front of it
    }
    DIAG_STAMP(NC - 1, 0);
    mphase(P1, NC - 1, -1);
    DIAG_STAMP(NC - 1, 1);
  }
#undef CORR_STEP
#undef MAIN_STEP
#undef LDS_WAIT
#undef LDS_RD
  __syncthreads();
  DIAG_STAMP(NC - 1, 3);
  DIAG_STAMP(30, 3);   // K loop done
  DIAG_RSTAMP(31, 2);

  // ---- epilogue ---- accumulator layout of a 16x16 tile: column = lane & 15, row = 4 (lane >> 4) + register
  constexpr float kAcc = kNegLog2e / kF16WScale;   // the weight fragments carry 2^8
  if constexpr (diag::kC8NoEpi) {
    float v = 0.f;
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) v += acc[rb][cb][0] + acc[rb][cb][1] + acc[rb][cb][2] + acc[rb][cb][3];
    if (v == 123.456f) p.agg_x[0] = v;
    return;
  }
  if constexpr (!IS_M) {
    // s[row] = [b3] + sum_n w3[n] * SiLU(a2[row][n] + b2[n]) over this workgroup's 512 columns (edge_x_m16.hip)
    float part[32];
#pragma unroll
    for (int v = 0; v < 32; ++v) part[v] = 0.f;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int n = 16 * (cb0 + cb) + r15;
      const float bb = p.b2x[n], w = p.w3x[n];
      const f32x2 bb2 = {bb, bb}, w2 = {w, w}, k2 = {kAcc, kAcc}, one2 = {1.0f, 1.0f};
#pragma unroll
      for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          const f32x2 a2 = {acc[rb][cb][i], acc[rb][cb][i + 1]};
          const f32x2 t = __builtin_elementwise_fma(a2, k2, bb2);
          const f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
          const f32x2 d = e + one2;
          const f32x2 rr = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
          const f32x2 sv = t * rr;
          f32x2 pp = {part[rb * 4 + i], part[rb * 4 + i + 1]};
          pp = __builtin_elementwise_fma(w2, sv, pp);
          part[rb * 4 + i] = pp.x;
          part[rb * 4 + i + 1] = pp.y;
        }
    }
    {
      float t0, t1;
      butterfly16(part, lane, t0, t1);   // value indices 2 m, 2 m + 1 (m = lane & 15): row block m >> 1, register 2 (m & 1) + {0, 1}
      const int row = 16 * (r15 >> 1) + 4 * q4 + 2 * (r15 & 1);
      L.part[wave * kR + row] = t0;
      L.part[wave * kR + row + 1] = t1;
    }
    __syncthreads();
    if (tid < kR) {
      float v = half == 0 ? p.scal[0] : 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += L.part[w * kR + tid];
      L.val[tid] = v;
    }
    __syncthreads();
    coordinate_segment_sums(p, L, S, tile, half, tid, lane, wave);
  } else {
    // m = SiLU(a2 + b2), gate = sigmoid(wa . m + ba), messages m * gate summed per receiving node (:57-61); the form of
    // tile128::message_epilogue on 16x16 accumulator tiles: 2 column blocks per wave, 64 message values per lane
    float mval[8][CB][4];
    {
      float zp[32];
#pragma unroll
      for (int v = 0; v < 32; ++v) zp[v] = 0.f;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int n = 16 * (cb0 + cb) + r15;
        const float bb = p.b2m[n], wa = p.wa[n];
#pragma unroll
        for (int rb = 0; rb < 8; ++rb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float m = silu_s(fmaf(acc[rb][cb][i], kAcc, bb));   // = -log2(e) * m
            mval[rb][cb][i] = m;
            zp[rb * 4 + i] = fmaf(wa, m, zp[rb * 4 + i]);
          }
      }
      float t0, t1;
      butterfly16(zp, lane, t0, t1);
      const int row = 16 * (r15 >> 1) + 4 * q4 + 2 * (r15 & 1);
      L.part[wave * kR + row] = t0;
      L.part[wave * kR + row + 1] = t1;
    }
    __syncthreads();
    if (tid < kR) {
      float g = p.scal[1];
#pragma unroll
      for (int w = 0; w < 8; ++w) g += L.part[w * kR + tid];
      L.val[tid] = sigmoid_f(g) * kNegInvLog2e;   // also undoes the scale of mval
    }
    __syncthreads();
    for (int base = 0; base < S; base += kSegFast) {
      const int ns = min(kSegFast, S - base);
      if (base > 0) __syncthreads();   // the previous pass has been read
      for (int t = tid; t < ns * kR; t += kT) {
        const int seg = base + (t >> 7), row = t & 127;
        L.gseg[t] = (L.seg_of_row[row] == seg) ? L.val[row] : 0.f;
      }
      __syncthreads();
      for (int sg = 0; sg < ns; ++sg) {
        const int seg = base + sg;
        const float* gw = L.gseg + sg * kR + 4 * q4;
        float v[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) v[cb] = 0.f;
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(gw + 16 * rb);   // rows 16 rb + 4 q4 + {0..3}
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[cb] = fmaf(mval[rb][cb][i], g4[i], v[cb]);
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          v[cb] += __shfl_xor(v[cb], 16);
          v[cb] += __shfl_xor(v[cb], 32);
        }
        if (q4 == 0) {
          const int mode = L.seg_mode[seg];
          float* dstp = mode == 2 ? p.agg_m + (size_t)L.seg_node[seg] * p.MP : p.part_m + ((size_t)tile * 2 + mode) * p.MP;
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) dstp[16 * (cb0 + cb) + r15] = v[cb];
        }
      }
    }
  }
  DIAG_STAMP(31, 0);   // epilogue done
}

// e4m3 B fragments of the correction product for v_mfma_scale_f32_16x16x128_f8f6f4, one instruction per 64-deep chunk:
//   out[((nb * NC + c) * 2 + piece) * 1024 + lane * 16 + j],  lane l: column 16 nb + (l & 15), K block q = l >> 4
//   block q covers hidden units 64 c + 32 (q >> 1) + [0, 32): even q holds e4m3(2^s_hi W_hi), odd q e4m3(2^s_lo W_lo) with
//   v = W * scale (what the fp16 stream holds), W_hi = fp16(v), W_lo = v - W_hi;
//   register piece `piece` of the lane holds hidden units 16 piece .. + 15 of its block (the A operand's 32 bytes in address order).
// The scale exponents come from the matrix's largest |v| (c8_absmax_kernel): 2^s_hi max in [112, 224] (e4m3 tops out at 448),
// s_lo = s_hi + 11 (a remainder is at most 2^-11 of its head); exps[0..1] = {127 - s_hi, 127 - s_lo} = the e8m0 bytes of the block
// scales, written by block 0 for the edge kernels.
__device__ __forceinline__ int c8_shift(unsigned maxbits) {
  const float mx = fminf(__builtin_bit_cast(float, maxbits), 65504.f);
  int s_hi = 0;
  if (mx > 0.f) s_hi = (int)floorf(log2f(224.0f / mx));
  return s_hi > 40 ? 40 : (s_hi < -40 ? -40 : s_hi);
}
__global__ void pack_frags_c8(const float* __restrict__ W, int Nout, int K, int ldw, int NP, int KP, unsigned char* __restrict__ out,
                              float scale, const unsigned* __restrict__ maxbits, int* __restrict__ exps) {
  __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);   // MODE.FP16_OVFL: the conversions saturate
  const int NC = KP / 64;
  const int s_hi = c8_shift(*maxbits);
  if (blockIdx.x == 0 && threadIdx.x == 0) { exps[0] = 127 - s_hi; exps[1] = 127 - s_hi - 11; }
  const float inv_hi = __builtin_ldexpf(1.0f, -s_hi), inv_lo = __builtin_ldexpf(1.0f, -s_hi - 11);   // 2^-s: the cvt divides
  const size_t total = (size_t)(NP / 16) * NC * 2 * 64 * 8;   // byte PAIRS
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int jp = i & 7, lane = (i >> 3) & 63, piece = (i >> 9) & 1;
    const size_t f = i >> 10;
    const int c = f % NC, nb = f / NC;
    const int q = lane >> 4, n = 16 * nb + (lane & 15);
    const int k = 64 * c + 32 * (q >> 1) + 16 * piece + 2 * jp;
    float v[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float w = (n < Nout && k + t < K) ? W[(size_t)n * ldw + k + t] * scale : 0.f;
      w = fminf(fmaxf(w, -65504.f), 65504.f);
      const float hi = (float)(_Float16)w;
      v[t] = (q & 1) ? w - hi : hi;
    }
    typedef __attribute__((ext_vector_type(2))) short i16x2_;
    i16x2_ r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, v[0], v[1], (q & 1) ? inv_lo : inv_hi, false);
    reinterpret_cast<unsigned short*>(out)[i] = (unsigned short)r.x;
  }
}
// largest |W scale| of the matrix as float bits (non-negative floats order like unsigned integers): grid-stride maximum, one
// atomicMax per workgroup into *maxbits (zeroed by the caller).  (The first build reduced in ONE workgroup: 0.87 ms per matrix,
// 7 ms of every TRAINING step, which repacks all layers -- profiles/r05u_train_f16c8_summary.txt.)
__global__ void c8_absmax_kernel(const float* __restrict__ W, int Nout, int K, int ldw, float scale, unsigned* __restrict__ maxbits) {
  __shared__ float red[256];
  float m = 0.f;
  const size_t total = (size_t)Nout * K;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float v = fabsf(W[(i / K) * ldw + (i % K)] * scale);
    m = v > m ? v : m;   // (a NaN weight never becomes the maximum: the scale stays finite)
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(maxbits, __builtin_bit_cast(unsigned, red[0]));
}

}  // namespace

int edge_f16c8_x_split(int WxP) { return WxP / (128 * kCBX); }   // column shares of the coordinate sums node_post adds

int init_edge_f16c8_attributes() {
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_c8_kernel<false, kCBX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_c8_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return EGNN_OK;
}

// the shapes of edge_x_m16.hip (hidden width 512 / 1024, 256 message columns); fp16 16-column streams, e4m3 streams and their
// scale exponents packed; fp32 table rows addressed by a 32-bit buffer offset
bool edge_f16c8_supported(const EdgeParams& p) {
  return (p.WxP == 512 || p.WxP == 1024) && p.MP == 256 && p.WmP % 64 == 0 && p.WmP >= 128 && p.w2x16 && p.w2m16 && p.w2x_c8 && p.w2m_c8 &&
         p.c8_exp && c8_smem_bytes(p.WxP, false) <= 160 * 1024 && c8_smem_bytes(p.WmP, true) <= 160 * 1024 &&
         (size_t)p.N * p.TC * 4 < ((size_t)1 << 32);
}

// coordinate kernel (WxP / 512 column shares per tile), then the message kernel (on `side` when the caller forks)
int launch_edge_f16c8_x(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR - 1) / kR;
  hipLaunchKernelGGL((edge_c8_kernel<false, kCBX>), dim3(tiles * (p.WxP / (128 * kCBX))), dim3(kT), c8_smem_bytes(p.WxP, false), st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}
int launch_edge_f16c8_m(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR - 1) / kR;
  hipLaunchKernelGGL((edge_c8_kernel<true, 2>), dim3(tiles), dim3(kT), c8_smem_bytes(p.WmP, true), st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

// out: e4m3 fragment stream of NP x KP bytes x 2; exps: int[2] (e8m0 bytes of the block scales); maxbits: one scratch word
int pack_c8_stream(const float* W, int Nout, int K, int ldw, int NP, int KP, void* out, float scale, int* exps, unsigned* maxbits,
                   hipStream_t st) {
  EGNN_HIP(hipMemsetAsync(maxbits, 0, sizeof(unsigned), st));
  hipLaunchKernelGGL(c8_absmax_kernel, dim3(128), dim3(256), 0, st, W, Nout, K, ldw, scale, maxbits);
  hipLaunchKernelGGL(pack_frags_c8, dim3(256), dim3(256), 0, st, W, Nout, K, ldw, NP, KP, static_cast<unsigned char*>(out), scale,
                     maxbits, exps);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace egnn
