// Per-graph fused dgrad, TWO workgroups per CU (round 4, second form; edge_bwd_dgrad_graph.hip is the first: read its header for
// what the kernel computes and why the fusion fits).  The first form needs 157 KB of LDS, so one workgroup owns a CU and its
// SiLU' epilogue (0.8 of 3.05 ms per launch for mlp_x, 0.73 of 1.48 ms for mlp_m) runs with the matrix cores idle -- both
// waves of a SIMD are always in the same phase.  Here a workgroup is 4 waves on a 128-column slice and fits twice on a CU
// (75 KB each, 2 x 4 waves of <= 256 registers), so one workgroup's epilogue runs under the other's K loop:
//   * first-layer table slice: 128 columns, 35 KB;
//   * operand ring 2 deep (33 KB): chunk c + 1 is written while chunk c is multiplied, one barrier per chunk;
//   * the one-hot incidence images as BIT masks (2 KB instead of 35 KB): [D | S][64 nodes][8 groups of 16 rows] 16-bit words
//     set with ds_or_b32 by the tile's prologue; a lane fetches its node's 8 words with one 16-byte read and turns two nibbles
//     into an MFMA A fragment through a 16-entry LDS table (nibble -> four bf16 of 0 / 1).
// Weight traffic from L2 is unchanged (128 rows per tile; 8 slices of 128 columns instead of 4 of 256), the dL/da2 tile is read
// by 8 workgroups instead of 4 (same XCD, L2 hits).
#include <stdlib.h>

#include "diag.h"
#include "kernels.h"

namespace egnn {
namespace {

constexpr int kT2 = 256, kR2 = 128, kRB2 = 4, kRPAD2 = kR2 + 1, kKC2 = 64, kNodes2 = 64, kCols2 = 128;
constexpr size_t kA12 = (size_t)8 * kRPAD2 * 16;   // one A chunk [8 k-groups][129][8 bf16]
constexpr int kTab2 = kCols2 * 2 + 16;             // bytes per node of the staged table slice (+16: rows 4 apart on different banks)
// LDS carve (byte offsets)
constexpr size_t k2OffPo = 0;                                  // int[R]   byte offset of the receiver's P row (row 64 = "no edge")
constexpr size_t k2OffQo = k2OffPo + kR2 * 4;                  // int[R]   byte offset of the sender's Q row
constexpr size_t k2OffD2 = k2OffQo + kR2 * 4;                  // float[R]
constexpr size_t k2OffWd = k2OffD2 + kR2 * 4;                  // float[128] scaled d^2 column of this slice
constexpr size_t k2OffX = k2OffWd + kCols2 * 4;                // float[64][3]
constexpr size_t k2OffPart = k2OffX + kNodes2 * 3 * 4;         // float[4][R] per-wave row sums
constexpr size_t k2OffLut = k2OffPart + 4 * kR2 * 4;           // u32[16][2]: nibble -> four bf16 (0 / 1)
constexpr size_t k2OffBits = k2OffLut + 16 * 8;                // u16[2][64 nodes][8 groups]: receiver / sender incidence bits of the tile
constexpr size_t k2OffP = k2OffBits + 2 * kNodes2 * 16;        // fp16[65][128 (+8)]
constexpr size_t k2OffQ = k2OffP + (size_t)(kNodes2 + 1) * kTab2;
constexpr size_t k2OffA1 = k2OffQ + (size_t)kNodes2 * kTab2;   // ring, 2 deep
constexpr size_t kSmem2 = k2OffA1 + 2 * kA12;
static_assert(kSmem2 <= 80 * 1024, "two workgroups per CU");
static_assert(k2OffLut % 16 == 0 && k2OffBits % 16 == 0 && k2OffP % 16 == 0 && k2OffA1 % 16 == 0, "16-byte LDS accesses");

struct DgradGraph2Params {
  int N, B;
  const int *graph_ptr, *row_ptr, *edge_dst, *edge_src;
  int e_base, n_edges;
  const float* x;
  const void* table;
  int TC, offP, offQ;
  const float* wd;
  const void* g_a2;
  int Kd;
  const void* w2t;
  int KP;
  __bf16 *Gd, *Gs;
  int ldg;
  float* cd;
  float* gd2_part;                 // [KP / 128][n_edges]
};

__global__ __launch_bounds__(kT2, 2) void edge_dgrad_graph2_kernel(const DgradGraph2Params p) {
  constexpr int NW = 4, PP = 4, NSET = 2;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* const s_po = reinterpret_cast<int*>(smem + k2OffPo);
  int* const s_qo = reinterpret_cast<int*>(smem + k2OffQo);
  float* const s_d2 = reinterpret_cast<float*>(smem + k2OffD2);
  float* const s_wd = reinterpret_cast<float*>(smem + k2OffWd);
  float* const s_x = reinterpret_cast<float*>(smem + k2OffX);
  float* const s_part = reinterpret_cast<float*>(smem + k2OffPart);
  unsigned* const s_bits32 = reinterpret_cast<unsigned*>(smem + k2OffBits);
  char* const s_a1 = smem + k2OffA1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int nslice = p.KP / kCols2;
  const int j = xcd_tile(blockIdx.x, gridDim.x);   // the column slices of a graph read the same dL/da2 rows: same XCD (one L2)
  const int g = j / nslice, slice = j - g * nslice;
  const int n0 = p.graph_ptr[g], nn = p.graph_ptr[g + 1] - n0;
  const int e_lo = p.row_ptr[n0], e_hi = p.row_ptr[n0 + nn];
  if (e_hi <= e_lo || e_lo < p.e_base || e_hi > p.e_base + p.n_edges || nn > kNodes2) return;   // (uniform)

  // ---- once per workgroup ----
  {
    const char* tab = static_cast<const char*>(p.table);
    for (int i = tid; i < kNodes2 * 16 * 2; i += kT2) {   // 16-byte pieces: [P | Q][64 nodes][16 pieces]
      const int which = i / (kNodes2 * 16), rem = i - which * (kNodes2 * 16), node = rem >> 4, piece = rem & 15;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (node < nn)
        v = *reinterpret_cast<const u32x4*>(tab + ((size_t)(n0 + node) * p.TC + (which ? p.offQ : p.offP) + slice * kCols2 + piece * 8) * 2);
      *reinterpret_cast<u32x4*>(smem + ((which ? (int)k2OffQ : (int)k2OffP) + node * kTab2 + piece * 16)) = v;
    }
    if (tid < kCols2) reinterpret_cast<_Float16*>(smem + ((int)k2OffP + kNodes2 * kTab2))[tid] = (_Float16)60000.0f;   // "no edge" row
    if (tid < 2 * kNodes2 * 16 / 16) reinterpret_cast<u32x4*>(smem + k2OffBits)[tid] = u32x4{0u, 0u, 0u, 0u};
    if (tid < 16) {   // nibble b3 b2 b1 b0 -> {bf16(b0) | bf16(b1) << 16, bf16(b2) | bf16(b3) << 16}
      const unsigned lo = ((tid & 1) | ((tid & 2) << 15)) * 0x3F80u, hi = (((tid >> 2) & 1) | (((tid >> 2) & 2) << 15)) * 0x3F80u;
      reinterpret_cast<u32x2*>(smem + k2OffLut)[tid] = u32x2{lo, hi};
    }
    for (int i = tid; i < nn * 3; i += kT2) s_x[i] = p.x[(size_t)3 * n0 + i];
    if (tid < kCols2) s_wd[tid] = p.wd[slice * kCols2 + tid];
  }
  f32x16 gd[2], gs[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) { gd[mb][i] = 0.f; gs[mb][i] = 0.f; }
  f32x2 cd2 = {0.f, 0.f};
  const int col = 32 * wave + r;
  const unsigned colb = 2u * (unsigned)col;
  const int NC = diag::kDgNoK ? 2 : p.Kd / kKC2, KS = p.Kd / 16;
  const int brow = tid >> 3, kg = tid & 7;            // this thread copies rows brow + 32 i (i < 4), k-group kg of every chunk
  const rsrc_t rs_g = make_rsrc(p.g_a2, (unsigned)((size_t)p.n_edges * p.Kd * 2));
  const rsrc_t rs_w = make_rsrc(p.w2t, (unsigned)((size_t)p.KP * p.Kd * 2));
  const unsigned vstep = (unsigned)(8 * NW) * (unsigned)p.Kd * 2u;
  char* const slot0 = s_a1 + (kg * kRPAD2 + brow) * 16;
  constexpr unsigned kSlotStep = 8 * NW * 16;
  const unsigned lane16 = lane * 16u;
  const unsigned lds_a1_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_a1 + (hh * kRPAD2 + r) * 16);
  const int colblk0 = slice * NW + wave;
  const unsigned w0off = (unsigned)colblk0 * KS * 1024u;
  float* const part_out = p.gd2_part + (size_t)slice * p.n_edges;
  const char* const bits_rd = smem + ((int)k2OffBits + r * 16);   // this lane's node (+ 32 mb) words: 8 groups x 16 bits

  const int ntiles = (e_hi - e_lo + kR2 - 1) / kR2;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int e0 = e_lo + tile * kR2;
    const int nvalid = min(kR2, e_hi - e0);
    __syncthreads();   // the previous tile's bits are cleared and its row arrays read; (first tile) the staged data are complete
    if (tid < kR2) {
      int dl = kNodes2, sl = 0;
      float dd = 0.f;
      if (tid < nvalid) {
        dl = p.edge_dst[e0 + tid] - n0;
        sl = p.edge_src[e0 + tid] - n0;
        const float dx = s_x[3 * dl] - s_x[3 * sl], dy = s_x[3 * dl + 1] - s_x[3 * sl + 1], dz = s_x[3 * dl + 2] - s_x[3 * sl + 2];
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);   // norm(...)**2 as in the forward (:56)
        dd = nrm * nrm;
        // incidence bits: word (node, group of 16 rows), bit = row inside the group; two words per 32-bit LDS word
        const int grp = tid >> 4, bit = tid & 15;
        atomicOr(s_bits32 + (dl * 8 + grp) / 2, 1u << (16 * (grp & 1) + bit));
        atomicOr(s_bits32 + (kNodes2 * 8 + sl * 8 + grp) / 2, 1u << (16 * (grp & 1) + bit));
      }
      s_po[tid] = (int)k2OffP + dl * kTab2;
      s_qo[tid] = (int)k2OffQ + sl * kTab2;
      s_d2[tid] = dd;
    }
    // ---- K loop: acc = dL/da2 tile . W2 slice; 2-deep ring ----
    f32x16 acc[kRB2];
#pragma unroll
    for (int rb = 0; rb < kRB2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;
    const unsigned vrow0 = (unsigned)(e0 - p.e_base + brow) * (unsigned)p.Kd * 2u + (unsigned)kg * 16u;
    auto gload = [&](const unsigned vrow, const int cq) {
      const int c = cq < NC ? cq : NC - 1;   // past the end: a harmless repeat of the last chunk
      return ldbuf_bf16x8(rs_g, vrow, (unsigned)c * kKC2 * 2u);
    };
    bf16x8 gsr[NSET][PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) gsr[0][i] = gload(vrow0 + i * vstep, 0);
#pragma unroll
    for (int i = 0; i < PP; ++i) *reinterpret_cast<bf16x8*>(slot0 + i * kSlotStep) = gsr[0][i];
#pragma unroll
    for (int q = 0; q < NSET; ++q)
#pragma unroll
      for (int i = 0; i < PP; ++i) gsr[q][i] = gload(vrow0 + i * vstep, 1 + q);   // set q: chunk 1 + q
    bf16x8 bq[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bq[s] = ldbuf_bf16x8(rs_w, lane16, w0off + (unsigned)s * 1024u);
    __syncthreads();

    bf16x8 a[kRB2];
#define LDS_RD(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define LDS_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
    // chunk c: multiply it from buffer c & 1; `copy`: write chunk c + 1 (register set xs) into the other buffer and request
    // chunk c + 3 into the set.  No read of the next chunk before the barrier (it is being written).
    auto chunk = [&](const int c, const bool copy, const bool last, bf16x8 (&xs)[PP]) {
      const unsigned abase = lds_a1_base + (unsigned)(c & 1) * (unsigned)kA12;
      char* const wr = slot0 + ((c + 1) & 1) * kA12;
      LDS_RD(a[0], abase, 0); LDS_RD(a[1], abase, 512); LDS_RD(a[2], abase, 1024); LDS_RD(a[3], abase, 1536);
#define GROUP(S, RB)                                                                                          \
      {                                                                                                       \
        if ((S) < 3 || (RB) == 0) LDS_WAIT(3);                                                                \
        else if ((RB) == 1) LDS_WAIT(2);                                                                      \
        else if ((RB) == 2) LDS_WAIT(1);                                                                      \
        else LDS_WAIT(0);                                                                                     \
        asm volatile("" : "+v"(a[RB]));                                                                       \
        acc[RB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[RB], bq[S], acc[RB], 0, 0, 0);                    \
        if ((S) < 3) LDS_RD(a[RB], abase, ((S) + 1) * 4128 + (RB) * 512);                                     \
      }
#define KSTEP(S)                                                                                              \
      GROUP(S, 0) GROUP(S, 1) GROUP(S, 2) GROUP(S, 3)                                                         \
      if (copy) {                                                                                             \
        *reinterpret_cast<bf16x8*>(wr + (S) * kSlotStep) = xs[S];                                             \
        xs[S] = gload(vrow0 + (S) * vstep, c + 1 + NSET);                                                     \
      }                                                                                                       \
      if (!last) {                                                                                            \
        const unsigned ksn = (unsigned)((c + 1) * 4 + (S)) * 1024u;                                           \
        bq[S] = ldbuf_bf16x8(rs_w, lane16, w0off + ksn);                                                      \
      }
      KSTEP(0) KSTEP(1) KSTEP(2) KSTEP(3)
#undef KSTEP
#undef GROUP
    };
    {
      int c = 0;
      for (; c + NSET <= NC - 1; c += NSET) {
#pragma unroll
        for (int q = 0; q < NSET; ++q) { chunk(c + q, true, false, gsr[q]); __syncthreads(); }
      }
      if (c < NC - 1) { chunk(c, true, false, gsr[0]); __syncthreads(); ++c; }
      chunk(NC - 1, false, true, gsr[0]);
    }
#undef LDS_WAIT
#undef LDS_RD

    // ---- epilogue in accumulator layout: row = 32 rb + acc_row(i, lane), column = col ----
    if constexpr (diag::kDgNoEpi) {
#pragma unroll
      for (int rb = 0; rb < kRB2; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) cd2.x += acc[rb][i];
      __syncthreads();
      continue;
    }
    const float wdc = s_wd[col];
    const f32x2 one2 = {1.0f, 1.0f}, k2 = {kNegInvLog2e, kNegInvLog2e}, wd2 = {wdc, wdc};
    // this lane's nodes' incidence words: [D | S][mb] x 8 groups of 16 rows
    u32x4 bw[2][2];
#pragma unroll
    for (int ds_ = 0; ds_ < 2; ++ds_)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) bw[ds_][mb] = *reinterpret_cast<const u32x4*>(bits_rd + (ds_ * kNodes2 + 32 * mb) * 16);
    float rowdot[32];
#pragma unroll
    for (int rb = 0; rb < kRB2; ++rb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rbase = 32 * rb + 8 * q + 4 * hh;
        const i32x4 po = *reinterpret_cast<const i32x4*>(s_po + rbase), qo = *reinterpret_cast<const i32x4*>(s_qo + rbase);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(s_d2 + rbase);
#pragma unroll
        for (int jj = 0; jj < 4; jj += 2) {
          const f16x2 pv = {*reinterpret_cast<const _Float16*>(smem + (unsigned)po[jj] + colb),
                            *reinterpret_cast<const _Float16*>(smem + (unsigned)po[jj + 1] + colb)};
          const f16x2 qv = {*reinterpret_cast<const _Float16*>(smem + (unsigned)qo[jj] + colb),
                            *reinterpret_cast<const _Float16*>(smem + (unsigned)qo[jj + 1] + colb)};
          const f16x2 a1 = pv + qv;
          const f32x2 dd = {d4[jj], d4[jj + 1]};
          const f32x2 a1f = {(float)a1.x, (float)a1.y};
          const f32x2 t2 = __builtin_elementwise_fma(wd2, dd, a1f);
          const f32x2 e = {__builtin_amdgcn_exp2f(t2.x), __builtin_amdgcn_exp2f(t2.y)};
          const f32x2 den = e + one2;
          const f32x2 sg = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
          const f32x2 sv = (t2 * k2) * sg;
          const f32x2 ds = __builtin_elementwise_fma(sv, one2 - sg, sg);
          const f32x2 gg = {acc[rb][4 * q + jj], acc[rb][4 * q + jj + 1]};
          const f32x2 g1 = diag::kDggNoSilu ? gg : gg * ds;
          acc[rb][4 * q + jj] = g1.x;
          acc[rb][4 * q + jj + 1] = g1.y;
          cd2 = __builtin_elementwise_fma(g1, dd, cd2);
          const f32x2 rd = g1 * wd2;
          rowdot[(rb & 1) * 16 + 4 * q + jj] = rd.x;
          rowdot[(rb & 1) * 16 + 4 * q + jj + 1] = rd.y;
        }
      }
      bf16x8 hf[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) hf[i >> 3][i & 7] = (__bf16)acc[rb][i];
#pragma unroll
      for (int s = 0; s < (diag::kDggNoHot ? 0 : 2); ++s) {
        const int grp = 2 * rb + s;                       // group of 16 rows; word = 16 bits of dword grp / 2
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          // A fragment: rows 4 hh .. + 3 (k = 8 hh + 0 .. 3) and 8 + 4 hh .. + 3 (k = 8 hh + 4 .. 7) of the group
          auto frag = [&](const unsigned dword) {
            const unsigned w = dword >> (16 * (grp & 1));
            const unsigned na = (w >> (4 * hh)) & 15u, nb = (w >> (8 + 4 * hh)) & 15u;
            const u32x2 fa = *reinterpret_cast<const u32x2*>(smem + k2OffLut + na * 8), fb = *reinterpret_cast<const u32x2*>(smem + k2OffLut + nb * 8);
            return __builtin_bit_cast(bf16x8, u32x4{fa.x, fa.y, fb.x, fb.y});
          };
          const bf16x8 od = frag(bw[0][mb][grp >> 1]), os = frag(bw[1][mb][grp >> 1]);
          gd[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(od, hf[s], gd[mb], 0, 0, 0);
          gs[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(os, hf[s], gs[mb], 0, 0, 0);
        }
      }
      if ((rb & 1) && !diag::kDggNoRow) {
        const float t = butterfly32(rowdot, lane);
        const int row = 32 * (r >> 4) + acc_row(r & 15, lane);
        s_part[wave * kR2 + 32 * (rb - 1) + row] = t;
      }
    }
    asm volatile("" : "+v"(cd2));   // (consumed here: else the compiler sinks the chain of multiply-adds to the loop latch)
    __syncthreads();
    if (tid < nvalid) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += s_part[w * kR2 + tid];
      part_out[e0 - p.e_base + tid] = v * kNegInvLog2e;
    }
    if (tid < 2 * kNodes2 * 16 / 16) reinterpret_cast<u32x4*>(smem + k2OffBits)[tid] = u32x4{0u, 0u, 0u, 0u};   // clear the tile's bits
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int node = 32 * mb + acc_row(i, lane);
      if (node < nn) {
        const size_t o = (size_t)(n0 + node) * p.ldg + slice * kCols2 + col;
        p.Gd[o] = (__bf16)gd[mb][i];
        p.Gs[o] = (__bf16)gs[mb][i];
      }
    }
  float cdv = cd2.x + cd2.y;
  cdv += __shfl_xor(cdv, 32);
  if (hh == 0) p.cd[(size_t)g * p.KP + slice * kCols2 + col] = cdv;
}

}  // namespace

int init_edge_dgrad_graph2_attributes() {
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_dgrad_graph2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  return EGNN_OK;
}

int launch_edge_dgrad_graph2(int N, int B, const int* graph_ptr, const int* row_ptr, const int* dst, const int* src, int e_first,
                             int n_edges, const float* x, const void* table, int TC, int offP, int offQ, const float* wd,
                             const void* g_a2, int Kd, const void* w2t, int KP, void* Gd, void* Gs, int ldg, float* cd,
                             float* gd2_part, hipStream_t st) {
  if (Kd % 64 != 0 || Kd < 256 || KP % kCols2 != 0) { set_error("edge dgrad (graph form 2): unsupported widths Kd=%d KP=%d", Kd, KP); return EGNN_EINVAL; }
  if (((size_t)n_edges + kR2) * Kd * 2 >= ((size_t)1 << 32)) { set_error("edge dgrad (graph form 2): chunk too large"); return EGNN_EINVAL; }
  DgradGraph2Params p;
  p.N = N; p.B = B; p.graph_ptr = graph_ptr; p.row_ptr = row_ptr; p.edge_dst = dst; p.edge_src = src; p.e_base = e_first;
  p.n_edges = n_edges; p.x = x; p.table = table; p.TC = TC; p.offP = offP; p.offQ = offQ; p.wd = wd; p.g_a2 = g_a2; p.Kd = Kd;
  p.w2t = w2t; p.KP = KP; p.Gd = static_cast<__bf16*>(Gd); p.Gs = static_cast<__bf16*>(Gs); p.ldg = ldg; p.cd = cd; p.gd2_part = gd2_part;
  hipLaunchKernelGGL(edge_dgrad_graph2_kernel, dim3(B * (KP / kCols2)), dim3(kT2), kSmem2, st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace egnn
