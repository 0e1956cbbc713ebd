// Coordinate edge kernel on v_mfma_f32_16x16x32_bf16 (gfx950): the SAME workgroup tile as edge_kernel_bf16_v3<2, false>
// (128 edges x 512 columns of mlp_x.2, EquivariantGraphNeuralNetwork.py:19-25, :62-65; 8 waves, 128 rows x 64 columns per
// wave, 64-deep activation chunks, phase-opposed SIMD partners, weight fragments streamed from L2), with the matrix
// instruction's shape changed.  Why build it: the coordinate kernel runs against the power limit (held clock 1.9-2.1 GHz),
// and on random data the chip holds a higher clock on the 16x16x32 shape than on 32x32x16 at about equal cycles per FLOP
// (MI355X_MICROARCH.md "DVFS give-back" item 7, cdna_hip_programming.md rule 28: "build both at the same output tile per
// wave and keep the faster by wall").  What changes with the shape:
//   * operands: lane l holds A[row l & 15][k = 8 (l >> 4) + j], B[k = 8 (l >> 4) + j][col l & 15]: a k-step is 32 deep,
//     8 row blocks x 4 column blocks of 16 per wave, 32 MFMAs of 16 cycles per k-step, 128 accumulator registers as before;
//   * LDS activation image: [8 k-groups][128 rows][16 B] with the row's low three bits XORed by the k-group -- the 8 lanes
//     that write one row's eight 16-byte pieces hit eight different slots of the 128-B store bank row, and every 16-lane
//     group of an operand ds_read_b128 (rows 0-3 and 12-15 of one k-group + rows 4-11 of the next) reads 16 different
//     slots of the 256-B load bank row (no padding: 16 KiB per chunk);
//   * weight fragments packed [16-column block][32-deep k-step][lane][8 bf16] (pack_frags_bf16_n16);
//   * the row sums of the head s = w3 . SiLU(a2) + b3 run over the 16 lanes of a DPP row (tile128::butterfly16).
// Prologue and segment sums are the shared ones of edge_tile.h.
#include <cstdlib>

#include "diag.h"
#include "edge_tile.h"

namespace egnn {

namespace {

#ifdef EGNN_EXP_WGSTAMP   // diagnostic build (diag.h, tools/fwd_stamps.py)
__device__ unsigned long long g_xwg_stamps[20000][12];
#define WG_STAMP(k) DIAG_WG_STAMP(g_xwg_stamps, 20000, k)
#define WG_STAMP_HW() DIAG_WG_STAMP_HW(g_xwg_stamps, 20000, 5)
#else
#define WG_STAMP(k)
#define WG_STAMP_HW()
#endif

using namespace tile128;
constexpr int kT = 512;
constexpr int kKC = 64;                       // activation chunk depth
constexpr size_t kA1 = (size_t)8 * kR * 16;   // one activation chunk: [8 k-groups][128 rows][8 bf16], XOR-swizzled rows
__host__ __device__ inline size_t x16_smem_bytes(int KP) { return kOffLoop + 2 * kA1 + (size_t)KP * 4; }

typedef __attribute__((ext_vector_type(4))) float f32x4v;

// SAVE = true: the training forward (egcl_forward_save): the same kernel, which also leaves in HBM the activation chunks
// (s1_out, as the MFMA consumed them), the scaled second-layer pre-activations -log2(e) * (a2 + b2) (g_a2_out, row-major bf16
// through a per-wave LDS transpose) and this column share of s_e (s_half_out) -- what edge_kernel_bf16_v3<2, false, false, true>
// leaves, so that egcl_backward_heads_saved / the wgrad GEMMs find the same buffers.
// V8 = the MFMA operand type: bf16x8 (precision bf16) or f16x8 (precision fp16: v_mfma_f32_16x16x32_f16, same rate, 11
// significant bits; weights packed x 2^8, kernels.h "MFMA operand type").
// PERSIST = true (inference): one workgroup per CU walks over its share of the (tile, column share) units, and the NEXT unit's edge
// indices, coordinates, first table rows and first weight fragments are requested under the CURRENT unit's epilogue (4 us of vector
// arithmetic, the matrix pipe idle): the three dependent memory round trips of the prologue (2.2 us of a 26 us workgroup,
// profiles/r03t_fwd_wg_stamps.txt) and the dispatch gap between workgroups (0.3 us) leave the critical path.
template <bool SAVE, typename V8 = bf16x8, bool PERSIST = false>
__global__ __launch_bounds__(kT, 2) void edge_x_m16_kernel(const EdgeParams p) {
  static_assert(!(SAVE && OpTraits<V8>::f16), "the training forward keeps bf16 activations");
  static_assert(!(SAVE && PERSIST), "the persistent form is the inference kernel");
  if constexpr (OpTraits<V8>::f16) f16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lds L(smem);
  char* s_a1 = smem + kOffLoop;
  float* s_wd = reinterpret_cast<float*>(s_a1 + 2 * kA1);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r15 = lane & 15, q4 = lane >> 4;
  const int KP = p.WxP;
  const int nsplit = p.WxP / 512;
  const int NC = KP / kKC, KS = KP / 32;
  const int brow = tid >> 3, kg = tid & 7;   // this thread builds rows brow and brow + 64, columns 8 kg .. 8 kg + 7 of a chunk
  const rsrc_t rs_tab = make_rsrc(p.table, diag::drop_table_loads(p.dbg) ? 0u : (unsigned)((size_t)p.N * p.TC * 2));
  const rsrc_t rs_w = make_rsrc(p.w2x, diag::drop_weight_loads(p.dbg) ? 0u : (unsigned)((size_t)p.WxP * KP * 2));
  const unsigned offP = 0u, offQ = (unsigned)p.WxP * 2u;   // fp16 table: {Px | Qx | Pm | Qm}
  const unsigned lane16 = lane * 16u;
  const int nunits = PERSIST ? ((p.E + kR - 1) / kR) * nsplit : (int)gridDim.x;
  UnitH uc0, uc1;   // table pieces of chunk 0
  V8 bq[2][4];      // weight fragments of the 2 k-steps of the current chunk
  // PERSIST: what the prologue of the next unit needs, requested under the epilogue of the current one
  int pf_d0, pf_s0, pf_d1, pf_s1, pf_d, pf_s, pf_r0, pf_r1;   // (pf_r0/1: row_ptr[dst], row_ptr[dst + 1] of row tid: the segment modes)
  float pf_x[6];
  auto unit_of = [&](const int bv, int& tile_o, int& half_o) {
    const int jj = xcd_tile(bv, nunits);
    tile_o = jj / nsplit;
    half_o = jj - tile_o * nsplit;
  };
  // (tq: the caller's copy of the thread index -- an opaque one inside the unit loop, so that brow / kg are re-derived there instead of
  // living across the K loop)
  auto prefetch_indices = [&](const int bv, const int tq) {   // this thread's two build rows (all threads) and row tid (threads 0-127)
    int t, h;
    unit_of(bv, t, h);
    const int ne0 = t * kR, nnv = min(kR, p.E - ne0), br = tq >> 3;
    const bool r0 = br < nnv, r1 = br + 64 < nnv, rt = tq < nnv && tq < kR;
    pf_d0 = r0 ? p.edge_dst[ne0 + br] : 0; pf_s0 = r0 ? p.edge_src[ne0 + br] : 0;
    pf_d1 = r1 ? p.edge_dst[ne0 + br + 64] : 0; pf_s1 = r1 ? p.edge_src[ne0 + br + 64] : 0;
    pf_d = rt ? p.edge_dst[ne0 + tq] : 0; pf_s = rt ? p.edge_src[ne0 + tq] : 0;
  };
  auto prefetch_data = [&](const int bv, const int tq) {      // coordinates of row tid, table pieces of chunk 0, first weight fragments
    int t, h;
    unit_of(bv, t, h);
    const unsigned k16 = (unsigned)(tq & 7) * 16u, l16 = (unsigned)(tq & 63) * 16u;
#pragma unroll
    for (int k = 0; k < 3; ++k) { pf_x[k] = p.x[3 * pf_d + k]; pf_x[3 + k] = p.x[3 * pf_s + k]; }   // (threads >= 128: node 0, unused; written by every thread so that nothing is carried around the K loop)
    pf_r0 = p.row_ptr[pf_d]; pf_r1 = p.row_ptr[pf_d + 1];
    unith_load(uc0, rs_tab, (unsigned)pf_d0 * (unsigned)p.TC * 2u + k16, (unsigned)pf_s0 * (unsigned)p.TC * 2u + k16, offP, offQ);
    unith_load(uc1, rs_tab, (unsigned)pf_d1 * (unsigned)p.TC * 2u + k16, (unsigned)pf_s1 * (unsigned)p.TC * 2u + k16, offP, offQ);
    const unsigned nw0 = (unsigned)(h * 32 + wave * 4) * KS * 1024u;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) bq[s][cb] = ldbuf_v8<V8>(rs_w, l16, nw0 + ((unsigned)cb * KS + s) * 1024u);
  };
  if constexpr (PERSIST) {
    for (int i = tid; i < KP; i += kT) s_wd[i] = p.wdx[i];   // (ordered by the first unit's row barrier)
    prefetch_indices(blockIdx.x, tid);
    prefetch_data(blockIdx.x, tid);
  }
  int bv = blockIdx.x;
  do {
  const int j = xcd_tile(bv, nunits);
  const int tile = j / nsplit, half = j - tile * nsplit;
  const int e0 = tile * kR;
  const int nvalid = min(kR, p.E - e0);

  DIAG_STAMP_SETUP(p.stamps + (size_t)wave * 32 * 4);
  DIAG_STAMP(30, 0);   // kernel entry
  WG_STAMP(0);
  // PERSIST: the thread index as the prologue / the epilogue see it is re-derived per unit from an opaque copy -- otherwise hipcc hoists
  // their unit-invariant address arithmetic out of the unit loop and keeps it in registers ACROSS the K loop, whose build then runs
  // on four temporaries (serial exp -> add -> rcp -> mul chains: +4 % kernel time, profiles/r05U_xp_ab.txt)
  int tid_p = tid;
  if constexpr (PERSIST) asm volatile("" : "+v"(tid_p));
  const int lane_p = tid_p & 63;
  if constexpr (!PERSIST) {
    prologue_rows(p, L, e0, nvalid, p.wdx, KP, s_wd, tid);
  } else {   // prologue_rows of edge_tile.h from the prefetched registers
    const int tid = tid_p;
    if (tid < kR) {
      const float dx = pf_x[0] - pf_x[3], dy = pf_x[1] - pf_x[4], dz = pf_x[2] - pf_x[5];   // (rows past the list: index 0 twice = 0)
      L.dst[tid] = pf_d;
      L.src[tid] = pf_s;
      L.diff[tid] = dx; L.diff[kR + tid] = dy; L.diff[2 * kR + tid] = dz;
      const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
      L.d2[tid] = nrm * nrm;
      reinterpret_cast<int*>(L.gseg)[tid] = pf_r0;        // (the message kernels' gate scratch: free in this kernel)
      reinterpret_cast<int*>(L.gseg)[kR + tid] = pf_r1;
    }
    __syncthreads();
  }
  DIAG_STAMP(30, 1);   // edge rows and geometry ready
  WG_STAMP(1);

  // ---- K loop ----
  const unsigned vdst0 = (unsigned)(PERSIST ? pf_d0 : L.dst[brow]) * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const unsigned vsrc0 = (unsigned)(PERSIST ? pf_s0 : L.src[brow]) * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const unsigned vdst1 = (unsigned)(PERSIST ? pf_d1 : L.dst[brow + 64]) * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const unsigned vsrc1 = (unsigned)(PERSIST ? pf_s1 : L.src[brow + 64]) * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const float d2r0 = L.d2[brow], d2r1 = L.d2[brow + 64];
  char* slot0 = s_a1 + (size_t)kg * (kR * 16) + (size_t)(brow ^ kg) * 16;
  char* slot1 = slot0 + 64 * 16;
  // LDS byte address of this lane's operand piece in buffer 0 for the two k-steps of a chunk (k-groups q4 and 4 + q4)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)s_a1;
  const unsigned abase0 = lds0 + (unsigned)q4 * (kR * 16) + (unsigned)(r15 ^ q4) * 16u;
  const unsigned abase1 = lds0 + (unsigned)(4 + q4) * (kR * 16) + (unsigned)(r15 ^ (4 + q4)) * 16u;
  const int cb0 = half * 32 + wave * 4;       // first 16-column block of this wave
  const unsigned w0 = (unsigned)cb0 * KS * 1024u;

  f32x4v acc[8][4];
#pragma unroll
  for (int rb = 0; rb < 8; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[rb][cb][i] = 0.f;

  // training forward: the activation chunk also goes to HBM, 16 bytes per thread, 128 contiguous bytes per row and chunk
  // (both column shares build the same activations: only share 0 stores them)
  // (buffer stores: descriptor in scalar registers, one 32-bit offset per row, the chunk's column offset as the scalar offset;
  // rows past the edge list fall outside the descriptor and are dropped by the hardware.  Flat 64-bit addresses cost the
  // four registers that made hipcc spill 20 bytes per lane around the K loop.)
  const rsrc_t rs_s1 = make_rsrc(SAVE ? p.s1_out : nullptr, SAVE ? (unsigned)((size_t)p.E * KP * 2) : 0u);   // < 4 GiB: checked by the host
  const unsigned vs1_0 = (unsigned)(e0 + brow) * (unsigned)KP * 2u + (unsigned)kg * 16u, vs1_1 = vs1_0 + 64u * (unsigned)KP * 2u;
  auto s1_store = [&](const V8 o0, const V8 o1, const int c) {
    if (half != 0) return;
    if constexpr (diag::kNoS1) return;
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)c * kKC * 2u);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rs_s1, vs1_0, soff, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rs_s1, vs1_1, soff, 0);
  };
  // chunk 0: both rows' table pieces and the first weight fragments are requested, THEN the segment structure of the tile is
  // worked out (two barriers, waves 0 and 1 only) while they are in flight, then the activations are finished
  // (one row after the other and the segment structure in front of them: 2.3 us per 27 us workgroup, tools/fwd_stamps.py)
  if constexpr (!PERSIST) {
    unith_load(uc0, rs_tab, vdst0, vsrc0, offP, offQ);
    unith_load(uc1, rs_tab, vdst1, vsrc1, offP, offQ);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) bq[s][cb] = ldbuf_v8<V8>(rs_w, lane16, w0 + ((unsigned)cb * KS + s) * 1024u);
  }
  const int S = prologue_segments<false>(p, L, e0, nvalid, tid_p, lane_p, wave);   // seg_mode: under the first matrix phase, below
  if constexpr (PERSIST) {   // segment_mode of edge_tile.h from the prefetched row_ptr words of the segment's first row (read in the epilogue)
    if (tid_p < S) {
      const int* rp = reinterpret_cast<const int*>(L.gseg);
      const int r0 = L.seg_rs[tid_p];
      const bool first = (e0 + r0) == rp[r0], last = (e0 + L.seg_re[tid_p] + 1) == rp[kR + r0];
      L.seg_mode[tid_p] = (first && last) ? 2 : (first ? 1 : 0);
    }
  }
  {
    const V8 o0 = unith_finish<V8>(uc0, s_wd + kg * 8, d2r0, slot0);
    const V8 o1 = unith_finish<V8>(uc1, s_wd + kg * 8, d2r1, slot1);
    if constexpr (SAVE) s1_store(o0, o1, 0);
  }
  __syncthreads();

  // matrix phase of chunk c: 2 k-steps x (8 row blocks x 4 column blocks).  Operand pipeline as in edge_bf16_v3.hip: the
  // A pieces by inline-asm ds_read_b128 (hipcc sinks compiler-visible LDS reads to their use) through a ring of 3 register
  // sets, each refilled in place for the use 3 row blocks later right after its MFMAs were issued; LDS returns in order,
  // so lgkmcnt(2) before a use means "all but the 2 younger reads have landed".  The weight fragments of k-step s of chunk c + 1 are requested after the MFMAs of
  // k-step s of chunk c (a whole chunk of distance).
  auto mphase = [&](const int c, const bool last) {
    const unsigned boff = (unsigned)(c & 1) * (unsigned)kA1;
    const unsigned ab0 = abase0 + boff, ab1 = abase1 + boff;
    V8 a[3];   // ring of 3 operand pieces: use u = 8 s + rb takes a[u % 3], which is refilled for use u + 3 right after
                   // (a fourth set costs the 4 registers that made hipcc spill around the K loop: 63 MB of scratch traffic per launch)
#define LDS_RD(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define LDS_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
    LDS_RD(a[0], ab0, 0); LDS_RD(a[1], ab0, 256); LDS_RD(a[2], ab0, 512);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int rb = 0; rb < 8; ++rb) {
        const int u = 8 * s + rb;
        if (u <= 13) LDS_WAIT(2);
        else if (u == 14) LDS_WAIT(1);
        else LDS_WAIT(0);
        asm volatile("" : "+v"(a[u % 3]));   // uses of the piece stay below the wait
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[rb][cb] = mfma16(a[u % 3], bq[s][cb], acc[rb][cb]);
#ifdef XM16_PIN   // (A/B, tools/c8_ab2.sh: a scheduling barrier behind every group of MFMAs, as edge_f16c8.hip needs)
        __builtin_amdgcn_sched_barrier(0);
#endif
        // refill in place for use u + 3 (the MFMAs above have read the registers at issue): row block (u + 3) & 7 of k-step (u + 3) >> 3
        if (u == 0) LDS_RD(a[0], ab0, 768); if (u == 1) LDS_RD(a[1], ab0, 1024); if (u == 2) LDS_RD(a[2], ab0, 1280);
        if (u == 3) LDS_RD(a[0], ab0, 1536); if (u == 4) LDS_RD(a[1], ab0, 1792); if (u == 5) LDS_RD(a[2], ab1, 0);
        if (u == 6) LDS_RD(a[0], ab1, 256); if (u == 7) LDS_RD(a[1], ab1, 512); if (u == 8) LDS_RD(a[2], ab1, 768);
        if (u == 9) LDS_RD(a[0], ab1, 1024); if (u == 10) LDS_RD(a[1], ab1, 1280); if (u == 11) LDS_RD(a[2], ab1, 1536);
        if (u == 12) LDS_RD(a[0], ab1, 1792);
      }
      if (!last) {
        const unsigned ksn = (unsigned)((c + 1) * 2 + s) * 1024u;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) bq[s][cb] = ldbuf_v8<V8>(rs_w, lane16, w0 + (unsigned)cb * KS * 1024u + ksn);
      }
    }
#undef LDS_WAIT
#undef LDS_RD
  };
  UnitH ua0, ua1;
  auto vload = [&](const int cq) {   // table rows for the activations of chunk cq (clamped: a harmless repeat at the end)
    const int c = cq < NC ? cq : NC - 1;
    const unsigned kb = (unsigned)c * kKC * 2u;
    unith_load(ua0, rs_tab, vdst0, vsrc0, offP + kb, offQ + kb);
    unith_load(ua1, rs_tab, vdst1, vsrc1, offP + kb, offQ + kb);
  };
  auto vfinish = [&](const int c) {  // SiLU + bf16 pack of chunk c into its LDS buffer
    const size_t nbuf = (size_t)(c & 1) * kA1;
    __builtin_amdgcn_s_setprio(3);   // vector work wins issue arbitration over the partner wave's MFMAs
    const V8 o0 = unith_finish<V8>(ua0, s_wd + c * kKC + kg * 8, d2r0, slot0 + nbuf);
    const V8 o1 = unith_finish<V8>(ua1, s_wd + c * kKC + kg * 8, d2r1, slot1 + nbuf);
    __builtin_amdgcn_s_setprio(0);
    if constexpr (SAVE) s1_store(o0, o1, c);
  };
  // SIMD partners (waves w and w + 4) in opposite phase, one barrier per chunk (edge_bf16_v3.hip)
  DIAG_STAMP(30, 2);   // chunk 0 built, first weights requested
  WG_STAMP(2);
  DIAG_RSTAMP(31, 1);
  vload(1);
  if (wave < 4) {
    // row_ptr loads of the segment modes (needed by the epilogue only) ride under the first matrix phase of wave 0
    const int my_mode = (!PERSIST && tid_p < S) ? segment_mode(p, L, e0, tid_p) : 0;
    for (int i = 0; i < NC - 1; ++i) {
      mphase(i, false);
      if (!PERSIST && i == 0 && tid_p < S) L.seg_mode[tid_p] = my_mode;
      vfinish(i + 1);
      vload(i + 2);
      __syncthreads();
    }
  } else {
    for (int i = 0; i < NC - 1; ++i) {
      vfinish(i + 1);
      vload(i + 2);
      __builtin_amdgcn_sched_barrier(0);
      mphase(i, false);
      __syncthreads();
    }
  }
  mphase(NC - 1, true);
  __syncthreads();
  DIAG_STAMP(30, 3);   // K loop done
  WG_STAMP(3);
  DIAG_RSTAMP(31, 2);

  // ---- epilogue: s[row] = [b3] + sum_n w3[n] * SiLU(a2[row][n] + b2[n]) over this workgroup's 512 columns ----
  int tid_e = tid;
  if constexpr (PERSIST) asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, r15_e = lane_e & 15, q4_e = lane_e >> 4;
  // accumulator layout of the 16x16 tile: column = lane_e & 15, row = 4 (lane_e >> 4) + register
  const bool more = PERSIST && bv + (int)gridDim.x < nunits;   // (uniform)
  // (the last unit requests itself again: unconditional writes, so that no old value of the prefetch registers is carried around the K loop)
  const int bnext = more ? bv + (int)gridDim.x : bv;
  if constexpr (PERSIST) prefetch_indices(bnext, tid_e);
  float part[32];
#pragma unroll
  for (int v = 0; v < 32; ++v) part[v] = 0.f;
  if constexpr (SAVE && !diag::kNoStage) {   // the scaled pre-activations go to HBM first (the K-loop buffers are free behind the last barrier);
                          // the accumulators then hold them for the SiLU below
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const float bb = p.b2x[16 * (cb0 + cb) + r15_e];
#pragma unroll
      for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[rb][cb][i] = fmaf(acc[rb][cb][i], kNegLog2e, bb);
    }
    // Row-major bf16 through a per-wave LDS tile kept TRANSPOSED, T[64 columns][16 rows] (32 bytes per column): a lane_e's four
    // values of a column block are rows 4 q4_e .. 4 q4_e + 3 of ONE column = one 8-byte write (16-byte tile rows before: sixteen
    // 2-byte writes per row block), and ds_read_b64_tr_b16 hands lane_e j of a 16-lane_e group row j of four neighbouring columns
    // (the group addresses 4 tile rows x 16 elements; gemm_tn.hip reads its operands the same way).  Two such reads = 8 columns =
    // one 16-byte store; a store instruction covers 64 contiguous bytes of each of 16 rows.
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2s;
    char* tile = s_a1 + (size_t)wave * (64 * 32);
    const unsigned tile_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)tile;
    const int g16 = lane_e >> 4, j16 = lane_e & 15;
    const unsigned rd_addr = tile_lds + (unsigned)((8 * g16 + (j16 >> 2)) * 32 + (j16 & 3) * 8);   // column 8 g16 + (j >> 2), rows 4 (j & 3) ..
    __bf16* tout = static_cast<__bf16*>(p.g_a2_out) + (size_t)e0 * p.WxP + 16 * cb0;
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        u32x2s w;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w.x) : "v"(acc[rb][cb][0]), "v"(acc[rb][cb][1]));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w.y) : "v"(acc[rb][cb][2]), "v"(acc[rb][cb][3]));
        *reinterpret_cast<u32x2s*>(tile + (16 * cb + r15_e) * 32 + q4_e * 8) = w;
      }
      __builtin_amdgcn_wave_barrier();
      u32x2s t[2][2];   // [half of the 64 columns][4-column piece]
      asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                   "ds_read_b64_tr_b16 %0, %4 offset:0\n\t"
                   "ds_read_b64_tr_b16 %1, %4 offset:128\n\t"
                   "ds_read_b64_tr_b16 %2, %4 offset:1024\n\t"
                   "ds_read_b64_tr_b16 %3, %4 offset:1152\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(t[0][0]), "=&v"(t[0][1]), "=&v"(t[1][0]), "=&v"(t[1][1]) : "v"(rd_addr) : "memory");
      if (16 * rb + j16 < nvalid - (diag::kNoT2 ? 1000 : 0)) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const u32x4 o = {t[k][0].x, t[k][0].y, t[k][1].x, t[k][1].y};   // row 16 rb + j16, columns 32 k + 8 g16 .. + 7
          *reinterpret_cast<u32x4*>(tout + (size_t)(16 * rb + j16) * p.WxP + 32 * k + 8 * g16) = o;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // SiLU and the w3 product on register PAIRS (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two elements per issue slot; only
  // the transcendentals stay scalar).  Packed fp32 is an anti-lever beside MFMAs, but the epilogue has none.
  typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int n = 16 * (cb0 + cb) + r15_e;
    const float bb = p.b2x[n], w = p.w3x[n];
    constexpr float kAcc = kNegLog2e / OpTraits<V8>::wscale;   // fp16: the weight fragments carry 2^8
    const f32x2 bb2 = {bb, bb}, w2 = {w, w}, k2 = {kAcc, kAcc}, one2 = {1.0f, 1.0f};
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        const f32x2 a2 = {acc[rb][cb][i], acc[rb][cb][i + 1]};
        const f32x2 t = SAVE ? a2 : __builtin_elementwise_fma(a2, k2, bb2);
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
  WG_STAMP(6);   // second-layer SiLU + w3 products done
  if constexpr (PERSIST) prefetch_data(bnext, tid_e);   // (the accumulators are dead: registers to spare)
  {
    float t0, t1;
    butterfly16(part, lane_e, t0, t1);   // value indices 2 m, 2 m + 1 (m = lane_e & 15): row block m >> 1, register 2 (m & 1) + {0, 1}
    const int row = 16 * (r15_e >> 1) + 4 * q4_e + 2 * (r15_e & 1);
    L.part[wave * kR + row] = t0;
    L.part[wave * kR + row + 1] = t1;
  }
  __syncthreads();
  WG_STAMP(7);   // row partials of the 8 waves in LDS
  if (tid_e < kR) {
    float v = half == 0 ? p.scal[0] : 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += L.part[w * kR + tid_e];
    L.val[tid_e] = v;
    // training forward: this column share of s_e for the backward's dL/d(x_i - x_j) = dL/d(sum_x[i]) * s_e.  (Rounds 3 / 4 up to
    // r04m never stored it -- the comment above promised it, the statement was missing: egcl_backward_heads_saved then read
    // whatever the buffer held, zeros in a fresh process, i.e. the term was silently dropped; found by running the graph-form
    // gradient test after another training test, fixed with tests/test_training.py::test_kept_buffers_are_fully_written.)
    if constexpr (SAVE) { if (tid_e < nvalid && p.s_half_out) p.s_half_out[(size_t)half * p.E + e0 + tid_e] = v; }
  }
  __syncthreads();
  WG_STAMP(8);   // s_e per row ready
  coordinate_segment_sums(p, L, S, tile, half, tid_e, lane_e, wave);
  DIAG_STAMP(31, 0);   // epilogue done
  WG_STAMP(4);
  WG_STAMP_HW();
  if constexpr (!PERSIST) break;
  if (!more) break;
  bv += (int)gridDim.x;
  __syncthreads();   // the row / segment arrays of this unit have been read
  } while (true);
}

}  // namespace

int init_edge_x_m16_attributes() {
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_x_m16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_x_m16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_x_m16_kernel<false, f16x8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_x_m16_kernel<false, bf16x8, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_x_m16_kernel<false, f16x8, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
  return EGNN_OK;
}

// grid of the persistent form: one workgroup per CU when there are more than two units per CU, else 0 (EGNN_X_PERSIST=0: A/B switch,
// read per launch so that a test can compare the two forms bit for bit in one process)
static int persistent_grid(int units) {
  const char* e = getenv("EGNN_X_PERSIST");
  static int ncu = 0;
  if (e && atoi(e) == 0) return 0;
  if (ncu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    ncu = n;
  }
  return units > 2 * ncu ? ncu : 0;
}

bool edge_x_m16_supported(const EdgeParams& p) {
  return (p.WxP == 512 || p.WxP == 1024) && p.w2x16 != nullptr && x16_smem_bytes(p.WxP) <= 160 * 1024 &&
         (size_t)p.N * p.TC * 2 < ((size_t)1 << 32);
}

// coordinate kernel only; p.w2x16 = mlp_x.2 packed by pack_frags_bf16_n16 (scaled by -1/log2(e))
int launch_edge_x_m16(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR - 1) / kR;
  EdgeParams q = p;
  q.w2x = p.w2x16;
  const int units = tiles * (p.WxP / 512), pg = persistent_grid(units);
  if (pg > 0) hipLaunchKernelGGL((edge_x_m16_kernel<false, bf16x8, true>), dim3(pg), dim3(kT), x16_smem_bytes(p.WxP), st, q);
  else hipLaunchKernelGGL(edge_x_m16_kernel<false>, dim3(units), dim3(kT), x16_smem_bytes(p.WxP), st, q);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

// precision fp16: p.w2x16 = the fp16 fragment stream (pack_frags_n16<_Float16>, scaled by -2^8 / log2(e))
int launch_edge_x_m16_f16(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR - 1) / kR;
  EdgeParams q = p;
  q.w2x = p.w2x16;
  const int units = tiles * (p.WxP / 512), pg = persistent_grid(units);
  if (pg > 0) hipLaunchKernelGGL((edge_x_m16_kernel<false, f16x8, true>), dim3(pg), dim3(kT), x16_smem_bytes(p.WxP), st, q);
  else hipLaunchKernelGGL((edge_x_m16_kernel<false, f16x8>), dim3(units), dim3(kT), x16_smem_bytes(p.WxP), st, q);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

// training forward: p.s1_out / p.g_a2_out / p.s_half_out receive what the backward needs (see the kernel's SAVE mode)
int launch_edge_x_m16_save(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR - 1) / kR;
  EdgeParams q = p;
  q.w2x = p.w2x16;
  hipLaunchKernelGGL(edge_x_m16_kernel<true>, dim3(tiles * (p.WxP / 512)), dim3(kT), x16_smem_bytes(p.WxP), st, q);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace egnn

#ifdef EGNN_EXP_WGSTAMP
extern "C" int egnn_debug_xwg_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(egnn::g_xwg_stamps), sizeof(unsigned long long) * 20000 * 12) == hipSuccess ? 0 : -1;
}
#endif
