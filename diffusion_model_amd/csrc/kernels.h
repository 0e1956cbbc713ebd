// Device-side helpers shared by the kernel translation units (gfx950 only).
#pragma once
#include "common.h"

namespace egnn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float silu_f(float v) {
  // v * sigmoid(v); exp(-v) = inf for very negative v gives rcp = 0 and the correct limit -0
  return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}
// SiLU on a pre-scaled argument t = -log2(e) * z:  returns -log2(e) * silu(z)
__device__ __forceinline__ float silu_s(float t) { return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t)); }
constexpr float kNegLog2e = -1.4426950408889634f;
constexpr float kNegInvLog2e = -0.6931471805599453f;
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// row of accumulator register `reg` of a 32x32 MFMA tile for this lane (C/D layout, gfx950)
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }


// XCD-aware workgroup -> tile map: workgroups b and b+8 share an XCD (round-robin dispatch), so give
// each XCD a contiguous range of tiles; tiles of one graph then share an L2 (table rows are re-read
// by every tile of the graph).  Bijective for any grid size; affects speed only.
__device__ __forceinline__ int xcd_tile(int b, int nwg) {
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

struct EdgeParams {
  int N, E;
  const int* edge_dst;
  const int* edge_src;
  const int* row_ptr;
  const float* x;      // [N][3]
  const float* table;  // [N][TC]  (fp16 [N][TC] for the v3 path)
  int TC, WxP, WmP, MP, cbx, cbm;
  const float *wdx, *wdm, *b2x, *w3x, *b2m, *wa, *scal;
  const void *w2x, *w2m;
  const void *w2x_lo, *w2m_lo;   // bf16 remainders W - bf16(W) of the scaled second-layer weights, same fragment layout (bf16x3)
  const void* w2x16;   // mlp_x.2 as 16x16x32 B fragments (edge_x_m16.hip), scaled; null when not packed
  const void* w2m16;   // mlp_m.2 as 16x16x32 B fragments (edge_small.hip), scaled; null when not packed
  const void *w2x_c8, *w2m_c8;   // precision f16c8: e4m3 fragments of the correction product (edge_f16c8.hip: pack_frags_c8)
  const int* c8_exp;             // device int[4]: e8m0 bytes of the weight block scales {x: hi, lo, m: hi, lo}
  float *agg_m, *agg_x, *part_m, *part_x;
  size_t agg_x_stride, part_x_stride;  // elements between the column-split copies of agg_x / part_x
  unsigned long long* stamps;  // diagnostic builds only (EGNN_EXP_STAMP): s_memtime stamps of one workgroup
  int dbg;  // diagnostic builds only (diag.h, EGNN_DEBUG): bit0 drop weight loads, bit1 drop table loads; 0 and ignored otherwise
  // ---- backward recompute variants only (BWD = true instantiations; edges = a chunk [0, E) of the caller's edge list) ----
  const float* g_sum_x;   // [N][3]  dL/d(sum_x), already multiplied by 1/(G+1)
  const float* g_sum_m;   // [N][MP] dL/d(sum_m)
  void* s1_out;           // bf16 [E][KP]: first-layer activations as the MFMA consumed them (scaled by -log2(e))
  void* g_a2_out;         // bf16 [E][WxP] (coordinate kernel) / [E][MP] (message kernel): dL/d(second-layer pre-activation)
  float* s_half_out;      // coordinate kernel: [nsplit][E] this workgroup's share of s_e = w3 . SiLU(a2) (+ b3 in share 0)
  float *g_col_a, *g_col_b, *g_scalar;   // column sums: {g_b2x, g_w3, g_b3} / {g_b2m, g_wa, g_ba} (atomic adds)
};

// layout helper used by both host (size) and device (carve): ints/floats 12*R*4 bytes, then 2 A1
// buffers, then the message tile.
__host__ __device__ inline size_t edge_smem_small(int R) { return (size_t)12 * R * 4; }
__host__ __device__ inline size_t edge_a1_bytes(int R) {
  const size_t bf = (size_t)8 * (R + 1) * 16, f32 = (size_t)32 * R * 4;
  return (bf > f32 ? bf : f32);
}
__host__ __device__ inline size_t edge_smem_bytes(int R, int MP) {
  return edge_smem_small(R) + 2 * edge_a1_bytes(R) + (size_t)R * (MP + 1) * 4;
}


// Sum of d^2 over the edges node n receives, as left by the v4 edge kernels in component 3 of the coordinate sums
// (column-split copy 0): the node's own slot when all its edges sit in one tile, else its tile partials in tile order.
__device__ __forceinline__ float node_sq_sum(int n, const int* __restrict__ row_ptr, int R, const float* __restrict__ agg_x,
                                             const float* __restrict__ part_x) {
  const int rp0 = row_ptr[n], rp1 = row_ptr[n + 1];
  if (rp1 <= rp0) return 0.f;
  const int t0 = rp0 / R, t1 = (rp1 - 1) / R;
  if (t0 == t1) return agg_x[(size_t)n * 4 + 3];
  float v = part_x[((size_t)t0 * 2 + 1) * 4 + 3];
  for (int t = t0 + 1; t <= t1; ++t) v += part_x[((size_t)t * 2) * 4 + 3];
  return v;
}
// Sum over the nodes [lo, hi) of one graph by an aligned group of 8 lanes (g8 = lane's index in the group): every
// group that sums the same graph produces the same bits (fixed stride, fixed butterfly order).
// (graphs of <= 64 nodes: at most 8 nodes per lane, written as two unrolled passes so that the 8 CSR reads and then the
// 8 sum reads are in flight together -- at B = 1 nothing else hides their latency)
__device__ __forceinline__ float graph_sq_sum8(int lo, int hi, int g8, const int* __restrict__ row_ptr, int R,
                                               const float* __restrict__ agg_x, const float* __restrict__ part_x) {
  int rp0[8], rp1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int m = lo + g8 + 8 * k;
    const bool ok = m < hi;
    rp0[k] = ok ? row_ptr[m] : 0;
    rp1[k] = ok ? row_ptr[m + 1] : 0;
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int m = lo + g8 + 8 * k;
    float v = 0.f;
    if (rp1[k] > rp0[k]) {
      const int t0 = rp0[k] / R, t1 = (rp1[k] - 1) / R;
      if (t0 == t1) v = agg_x[(size_t)m * 4 + 3];
      else {
        v = part_x[((size_t)t0 * 2 + 1) * 4 + 3];
        for (int t = t0 + 1; t <= t1; ++t) v += part_x[((size_t)t * 2) * 4 + 3];
      }
    }
    s += v;
  }
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 4);
  return s;
}

// ---- helpers shared by the bf16 edge kernels ---------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
// soff must be wave-uniform; readfirstlane makes that provable to the compiler (else it wraps the load
// in a waterfall loop)
__device__ __forceinline__ f32x4 ldbuf_f32x4(rsrc_t rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, __builtin_amdgcn_readfirstlane(soff), 0));
}
__device__ __forceinline__ bf16x8 ldbuf_bf16x8(rsrc_t rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, __builtin_amdgcn_readfirstlane(soff), 0));
}

// ---- cross-lane row reduction without LDS ------------------------------------------------------------
// v[q], q = 0..31, per lane.  Returns the sum over the 32 lanes of this half-wave of value index
// q = (lane & 31).  Halving butterfly: every step pairs two lanes, each keeps one half of the value
// indices: v_permlane16_swap for lane^16, DPP row_ror:8 / row_half_mirror / quad_perm inside a row of
// 16, bank-masked DPP moves as the per-lane select.  ~80 VALU instructions, no LDS traffic.
// (The swap is inline asm: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane16_swap into
// one when they are summed as floats.  s_nop 1 = the 2 wait states between a VALU write and the swap.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, 0xF, 0xF, true));
}
template <int BANK>
__device__ __forceinline__ float dpp_sel(float a, float b) {  // lanes of the banks in BANK take b
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b), 0xE4, 0xF, BANK, false));
}
__device__ __forceinline__ float butterfly32(float (&v)[32], int lane) {
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    float a = v[q], b = v[q + 16];
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v[q] = a + b;   // even rows: index q over lanes {l, l+16}; odd rows: index q + 16
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float ta = v[q] + dpp_mov<0x128>(v[q]);          // row_ror:8  (lane ^ 8)
    const float tb = v[q + 8] + dpp_mov<0x128>(v[q + 8]);
    v[q] = dpp_sel<0xC>(ta, tb);                            // lanes with (lane & 8) keep index q + 8
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float ta = v[q] + dpp_mov<0x141>(v[q]);          // row_half_mirror (pairs lane l with 7 - l)
    const float tb = v[q + 4] + dpp_mov<0x141>(v[q + 4]);
    v[q] = dpp_sel<0xA>(ta, tb);                            // lanes with (lane & 4) keep index q + 4
  }
  const bool b2 = (lane & 2) != 0, b1 = (lane & 1) != 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float ta = v[q] + dpp_mov<0x4E>(v[q]);           // quad_perm [2,3,0,1]
    const float tb = v[q + 2] + dpp_mov<0x4E>(v[q + 2]);
    v[q] = b2 ? tb : ta;
  }
  const float ta = v[0] + dpp_mov<0xB1>(v[0]);             // quad_perm [1,0,3,2]
  const float tb = v[1] + dpp_mov<0xB1>(v[1]);
  return b1 ? tb : ta;
}

struct Unit {  // one build unit in flight: 8 columns of one row of one MLP
  f32x4 p0, p1, q0, q1;
};
// table rows through a buffer descriptor: voffset = byte offset of the row, soffset = column offset
__device__ __forceinline__ void unit_load(Unit& u, rsrc_t tab, unsigned vdst, unsigned vsrc, unsigned sP, unsigned sQ) {
  u.p0 = ldbuf_f32x4(tab, vdst, sP);
  u.p1 = ldbuf_f32x4(tab, vdst + 16, sP);
  u.q0 = ldbuf_f32x4(tab, vsrc, sQ);
  u.q1 = ldbuf_f32x4(tab, vsrc + 16, sQ);
}
__device__ __forceinline__ void unit_finish(const Unit& u, const float* wd, float d2, char* slot) {
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(wd), w1 = *reinterpret_cast<const f32x4*>(wd + 4);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[j] = (__bf16)silu_s(fmaf(w0[j], d2, u.p0[j] + u.q0[j]));   // table, wd pre-scaled by -log2(e)
    o[j + 4] = (__bf16)silu_s(fmaf(w1[j], d2, u.p1[j] + u.q1[j]));
  }
  *reinterpret_cast<bf16x8*>(slot) = o;
}


// ---- half-precision table units (edge_bf16_v3) ---------------------------------------------------------------
// The first-layer table of the v3 path is stored as fp16 (11 significant bits: finer than the bf16 the activation
// is rounded to afterwards): half the bytes through the vector-memory path, P + Q as packed-half adds and the
// fp16 -> fp32 conversion folded into the fma (v_fma_mix_f32).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- MFMA operand type of the half-precision paths: bf16x8 (precision bf16) or f16x8 (precision fp16) ------------------
// fp16 operands have 11 significant bits against bf16's 8 at the same matrix-core rate.  What the narrower exponent needs:
//  * the fp16 weight fragments are packed multiplied by kF16WScale = 2^8 (default-init and trained weights of a 1024-wide
//    layer are ~1e-2: a fraction of them would otherwise be fp16 subnormals) and the accumulators divided by it in the
//    epilogue's existing fma (the constant changes, no instruction is added);
//  * overflow saturates instead of producing inf (an infinite activation times a zero weight would be NaN where the bf16
//    path stays finite): the kernels set MODE.FP16_OVFL (f16_saturate_mode), which clamps an overflowing fp16 RESULT to
//    +-65504 in hardware; the packs clamp explicitly.
constexpr float kF16WScale = 256.0f;
template <typename V8> struct OpTraits;
template <> struct OpTraits<bf16x8> { typedef __bf16 elem; static constexpr bool f16 = false; static constexpr float wscale = 1.0f; };
template <> struct OpTraits<f16x8> { typedef _Float16 elem; static constexpr bool f16 = true; static constexpr float wscale = kF16WScale; };
__device__ __forceinline__ void f16_saturate_mode() {   // hwreg(HW_REG_MODE = 1, offset 23, size 1) = FP16_OVFL
  __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
}
template <typename V8>
__device__ __forceinline__ V8 pack8(const float (&v)[8]) {
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  const f32x8 f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  return __builtin_convertvector(f, V8);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <typename V8>
__device__ __forceinline__ V8 ldbuf_v8(rsrc_t rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, __builtin_amdgcn_readfirstlane(soff), 0));
}

struct UnitH {  // one build unit in flight: 8 columns of one row of one MLP
  f16x8 p, q;
};
__device__ __forceinline__ void unith_load(UnitH& u, rsrc_t tab, unsigned vdst, unsigned vsrc, unsigned sP, unsigned sQ) {
  u.p = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(tab, vdst, __builtin_amdgcn_readfirstlane(sP), 0));
  u.q = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(tab, vsrc, __builtin_amdgcn_readfirstlane(sQ), 0));
}
// V8 = bf16x8 (precision bf16) or f16x8 (precision fp16): the MFMA operand type the activation is rounded to
template <typename V8 = bf16x8>
__device__ __forceinline__ V8 unith_finish(const UnitH& u, const float* wd, float d2, char* slot) {
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(wd), w1 = *reinterpret_cast<const f32x4*>(wd + 4);
  const f16x8 t = u.p + u.q;
  typedef typename OpTraits<V8>::elem elem;
  V8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[j] = (elem)silu_s(fmaf(w0[j], d2, (float)t[j]));   // table, wd pre-scaled by -log2(e)
    o[j + 4] = (elem)silu_s(fmaf(w1[j], d2, (float)t[j + 4]));
  }
  *reinterpret_cast<V8*>(slot) = o;
  return o;
}

// ---- backward recompute: the second-layer nonlinearity and its derivative from the MFMA accumulator ----------------
// t2 = -log2(e) * a2 (scaled pre-activation incl. bias).  s = SiLU(a2), ds = SiLU'(a2) = sig + s * (1 - sig).
__device__ __forceinline__ void silu_grad_s(float t2, float& s, float& ds) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t2));
  s = (t2 * kNegInvLog2e) * sg;
  ds = fmaf(s, 1.0f - sg, sg);
}
// Store one wave's 32-row x 64-column block of an accumulator-layout tile as row-major bf16 with 16-byte accesses:
// v[cb][i] = value of (row acc_row(i, lane), column 32 cb + (lane & 31)); stg = this wave's [32][72] bf16 LDS scratch;
// out = address of (row 0, column 0) of the block in a row-major bf16 array with `ld` columns; rows >= nrows are skipped.
__device__ __forceinline__ void store_block_bf16(const f32x16 (&v)[2], int ncb, __bf16* stg, __bf16* out, size_t ld, int nrows,
                                                 int lane) {
  const int r = lane & 31;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
    if (cb < ncb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) stg[acc_row(i, lane) * 72 + 32 * cb + r] = (__bf16)v[cb][i];
    }
  __builtin_amdgcn_wave_barrier();
  const int pieces = 32 * (ncb * 4);   // 16-byte pieces: ncb*4 per row
  for (int q = lane; q < pieces; q += 64) {
    const int row = q / (ncb * 4), seg = q - row * (ncb * 4);
    if (row < nrows)
      *reinterpret_cast<bf16x8*>(out + (size_t)row * ld + 8 * seg) = *reinterpret_cast<const bf16x8*>(stg + row * 72 + 8 * seg);
  }
  __builtin_amdgcn_wave_barrier();
}

// node_post arguments (fp32 and bf16 variants)
struct PostParams {
  int N, H, MP, K1P, WhP, HP, R;
  const float *h, *x;
  const int *row_ptr, *node_graph;
  const float *agg_m, *agg_x, *part_m, *part_x, *gscale;
  const int* graph_ptr;   // [B+1]
  int sq_from_agg;        // 1: the per-graph sums of d^2 are taken from component 3 of the coordinate sums (v4 edge
                          //    kernels) by node_post itself; 0: gscale holds them
  size_t agg_x_stride, part_x_stride;
  int nsplit_x;   // column-split copies of the coordinate sums to add
  int per_graph;
  const f32x4 *w1h, *w2h;
  const void *w1h_bf16, *w2h_bf16p;   // bf16 node MLP (node_post_bf16_kernel)
  const void *w1h_lo = nullptr, *w2h_lo = nullptr;   // split-operand form: remainder streams (heads in w1h_bf16 / w2h_bf16p)
  int K1Q;                            // H + MP rounded up to 16
  const float *b1h, *b2h;
  float *h_out, *x_out;
  float* h_partial;   // [8][N][H] scratch of the hidden-split form (small N), or null
};
// hs_out: the hidden split that was launched (1 = none); defer_finish: leave the partial h' of a split launch to the caller
int launch_node_post_bf16(const PostParams& q, hipStream_t st, bool f16 = false, bool split = false, bool defer_finish = false,
                          int* hs_out = nullptr);
int launch_node_post_finish(int N, int H, int hs, const float* partial, const float* b2h, float* h_out, hipStream_t st);
bool node_post_split_supported(const PostParams& q);   // head + remainder fp16 operands, three MFMAs per product (node_bf16.hip)
int node_post_split_k();
bool node_post_bf16_supported(const PostParams& q);
int init_node_bf16_attributes();

bool edge_bf16_v3_supported(const EdgeParams& p);
int edge_v3_rows();
int launch_edge_bf16_v4_m(const EdgeParams& p, hipStream_t st);
int launch_edge_bf16_v3_x(const EdgeParams& p, hipStream_t st);
int launch_edge_bf16_v3_x_bwd(const EdgeParams& p, hipStream_t st);
int launch_edge_bf16_v4_m_bwd(const EdgeParams& p, hipStream_t st);
int launch_edge_bf16_v3_x_save(const EdgeParams& p, hipStream_t st);
int launch_edge_bf16_v4_m_save(const EdgeParams& p, hipStream_t st);
bool edge_bf16_v4_supported(const EdgeParams& p);
int edge_v4_rows();
int init_edge_bf16_v4_attributes();
int init_edge_bf16_v3_attributes();
int launch_edge_x_m16(const EdgeParams& p, hipStream_t st);   // coordinate kernel on v_mfma_f32_16x16x32_bf16
int launch_edge_x_m16_save(const EdgeParams& p, hipStream_t st);
int launch_edge_x_m16_f16(const EdgeParams& p, hipStream_t st);   // precision fp16: v_mfma_f32_16x16x32_f16 (p.w2x16 = fp16 fragments)
int launch_edge_f16_v4_m(const EdgeParams& p, hipStream_t st);    // precision fp16 message kernel (p.w2m = fp16 fragments)
bool edge_x_m16_supported(const EdgeParams& p);
int init_edge_x_m16_attributes();
// 32-edge tiles for small graphs (edge_small.hip): p.w2x16 / p.w2m16 = 16-column fragment streams of the operand type
int launch_edge_small_x(const EdgeParams& p, hipStream_t st, bool f16, int rows);   // rows = 32 or 64 edges per tile
int launch_edge_small_m(const EdgeParams& p, hipStream_t st, bool f16, int rows);
bool edge_small_supported(const EdgeParams& p);
int launch_edge_bf16x3(const EdgeParams& p, hipStream_t st);    // precision 'bf16x3': head / remainder split operands
bool edge_bf16x3_supported(const EdgeParams& p);
int init_edge_bf16x3_attributes();
// precision 'f16c8': fp16 main product + block-scaled e4m3 correction (edge_f16c8.hip); p.w2x16 / p.w2m16 = the fp16 16-column streams
int launch_edge_f16c8_x(const EdgeParams& p, hipStream_t st);
int launch_edge_f16c8_m(const EdgeParams& p, hipStream_t st);
bool edge_f16c8_supported(const EdgeParams& p);
int edge_f16c8_x_split(int WxP);
int init_edge_f16c8_attributes();
// the same precision on 32x32 matrix tiles (edge_f16c8w.hip); p.w2x / p.w2m = the fp16 32-column streams, p.w2x_c8 / p.w2m_c8 =
// the e4m3 streams of pack_c8w_stream (scale exponents: those pack_c8_stream wrote)
int launch_edge_f16c8w_x(const EdgeParams& p, hipStream_t st);
int launch_edge_f16c8w_m(const EdgeParams& p, hipStream_t st);
bool edge_f16c8w_supported(const EdgeParams& p);
int init_edge_f16c8w_attributes();
int pack_c8w_stream(const float* W, int Nout, int K, int ldw, int NP, int KP, void* out, float scale, const int* exps, hipStream_t st);
int pack_c8_stream(const float* W, int Nout, int K, int ldw, int NP, int KP, void* out, float scale, int* exps, unsigned* maxbits,
                   hipStream_t st);

}  // namespace egnn
