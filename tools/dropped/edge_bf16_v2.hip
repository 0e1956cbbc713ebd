// Fused per-edge kernel, bf16 MFMA throughput version (gfx950).
//
// One workgroup = 8 wave64 (two per SIMD) = one tile of 64 edges in CSR order.  Both edge MLPs of
// the layer (reference: EquivariantGraphNeuralNetwork.py:55-65) run in ONE K-loop over their hidden
// dimension, 64 columns per chunk:
//
//   build   512 threads = 64 rows x 8 column groups: a1 = SiLU(P[dst] + Q[src] + wd * d2) for the
//           coordinate MLP and for the message MLP, packed to bf16 straight into the LDS image that
//           the MFMA A-fragment reads expect (row-contiguous 16-byte slots: conflict-free b128).
//   MFMA    wave w owns columns [128w, 128w+128) of mlp_x.2 (W = 1024) and [32w, 32w+32) of mlp_m.2
//           (M = 256): per 16-deep k-step 2 A fragments per MLP from LDS, 4 + 1 B fragments straight
//           from the fragment-packed weights (one coalesced 1 KiB load each, L2 resident), 10 MFMAs.
//   The chunk c+1 activations are built while chunk c is multiplied: the table-row loads are issued a
//   half chunk ahead of their SiLU, the SiLUs sit between the MFMA groups, and the second wave of
//   each SIMD fills the matrix pipe while its partner does vector work.  One barrier per chunk.
//
// Nothing of size [E, *] is written to HBM: the [64, 1024] and [64, 256] second-layer outputs stay in
// accumulators and are reduced to s_ij (mlp_x.4) / gated messages, then segment-summed per receiving
// node exactly as in the fp32 kernel (bitwise deterministic, no atomics).
#include <type_traits>

#include "kernels.h"

namespace egnn {

namespace {

constexpr int kT2 = 512;  // threads
constexpr int kR = 64, kRB = 2, kRPAD = kR + 1, kKC = 64;
constexpr size_t kA1 = (size_t)8 * kRPAD * 16;  // bytes of one activation chunk [8 k-groups][65][8 bf16]

constexpr int kMaxSegFast = 8;  // segments per tile handled by the in-register segment sums
// dst, src, d2, diff[3], sval, gate, part[8], partg[8], seg tables[5] + misc, gseg[kMaxSegFast]
__host__ __device__ inline size_t v2_small_bytes() { return (size_t)(2 + 1 + 3 + 1 + 1 + 8 + 8 + 6 + kMaxSegFast) * kR * 4; }
__host__ __device__ inline size_t v2_smem_bytes(int MP) { return v2_small_bytes() + 4 * kA1 + (size_t)kR * (MP + 1) * 4; }

template <int CBX, int CBM>
__global__ __launch_bounds__(kT2, 2) void edge_kernel_bf16_v2(const EdgeParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_dst = reinterpret_cast<int*>(smem);
  int* s_src = s_dst + kR;
  float* s_d2 = reinterpret_cast<float*>(s_src + kR);
  float* s_diff = s_d2 + kR;  // [3][R]
  float* s_sval = s_diff + 3 * kR;
  float* s_gate = s_sval + kR;
  float* s_part = s_gate + kR;   // [8][R] mlp_x partial row sums per wave
  float* s_partg = s_part + 8 * kR;  // [8][R] gate partial row sums per wave
  int* s_seg_of_row = reinterpret_cast<int*>(s_partg + 8 * kR);
  int* s_seg_node = s_seg_of_row + kR;
  int* s_seg_rs = s_seg_node + kR;
  int* s_seg_re = s_seg_rs + kR;
  int* s_seg_mode = s_seg_re + kR;   // 2 = whole node -> agg, 1 / 0 = partial slot of this tile
  int* s_misc = s_seg_mode + kR;     // [0] = number of segments in the tile
  float* s_gseg = reinterpret_cast<float*>(s_misc + kR);  // [kMaxSegFast][R] gate * [row in segment]
  char* s_a1 = smem + v2_small_bytes();  // [x0, x1, m0, m1]
  float* s_msg = reinterpret_cast<float*>(s_a1 + 4 * kA1);
  // the d^2 weight columns live in the (not yet used) message tile during the K-loop
  float* s_wdx = s_msg;
  float* s_wdm = s_msg + p.WxP;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int e0 = tile * kR;
  const int nvalid = min(kR, p.E - e0);

  if (tid < kR) {
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
    s_diff[tid] = dx; s_diff[kR + tid] = dy; s_diff[2 * kR + tid] = dz;
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);  // norm(...)**2 as in the reference (:56)
    s_d2[tid] = nrm * nrm;
    // segments = runs of equal receiving node (CSR order); tid < 64 is exactly wave 0
    const int dprev = __shfl_up(d, 1), dnext = __shfl_down(d, 1);
    const bool valid = tid < nvalid;
    const bool is_start = valid && (tid == 0 || dprev != d);
    const bool is_end = valid && (tid == nvalid - 1 || dnext != d);
    const unsigned long long starts = __ballot(is_start);
    const int seg = __popcll(starts & ((2ull << tid) - 1ull)) - 1;
    s_seg_of_row[tid] = valid ? seg : -1;
    if (is_start) { s_seg_node[seg] = d; s_seg_rs[seg] = tid; }
    if (is_end) s_seg_re[seg] = tid;
    if (tid == 0) s_misc[0] = __popcll(starts);
  }
  for (int i = tid; i < p.WxP; i += kT2) { s_wdx[i] = p.wdx[i]; s_wdm[i] = p.wdm[i]; }
  __syncthreads();

  // ---- fused K-loop ----
  const int KP = p.WxP;            // == p.WmP (checked on the host)
  const int NC = KP / kKC, KS = KP / 16;
  const int brow = tid >> 3, kg = tid & 7;   // build unit of this thread
  const rsrc_t rs_tab = make_rsrc(p.table, (p.dbg & 2) ? 0u : (unsigned)((size_t)p.N * p.TC * 4));
  const rsrc_t rs_wx = make_rsrc(p.w2x, (p.dbg & 1) ? 0u : (unsigned)((size_t)KP * KP * 2));
  const rsrc_t rs_wm = make_rsrc(p.w2m, (p.dbg & 1) ? 0u : (unsigned)((size_t)p.MP * KP * 2));
  const unsigned vdst = (unsigned)s_dst[brow] * (unsigned)p.TC * 4u + (unsigned)kg * 32u;
  const unsigned vsrc = (unsigned)s_src[brow] * (unsigned)p.TC * 4u + (unsigned)kg * 32u;
  const float d2r = s_d2[brow];
  const unsigned offPx = 0, offQx = p.WxP * 4u, offPm = 2u * p.WxP * 4u, offQm = (2u * p.WxP + p.WmP) * 4u;
  char* slot_base = s_a1 + ((size_t)kg * kRPAD + brow) * 16;  // + buffer offset
  const unsigned lane16 = lane * 16u;
  const unsigned wx0 = (unsigned)(wave * CBX) * KS * 1024u, wm0 = (unsigned)(wave * CBM) * KS * 1024u;  // scalar

  f32x16 accx[kRB][CBX], accm[kRB][CBM];
#pragma unroll
  for (int rb = 0; rb < kRB; ++rb) {
#pragma unroll
    for (int cb = 0; cb < CBX; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) accx[rb][cb][i] = 0.f;
#pragma unroll
    for (int cb = 0; cb < CBM; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) accm[rb][cb][i] = 0.f;
  }

  {  // chunk 0
    Unit u;
    unit_load(u, rs_tab, vdst, vsrc, offPx, offQx);
    unit_finish(u, s_wdx + kg * 8, d2r, slot_base);
    unit_load(u, rs_tab, vdst, vsrc, offPm, offQm);
    unit_finish(u, s_wdm + kg * 8, d2r, slot_base + 2 * kA1);
  }
  bf16x8 bx[CBX], bm[CBM], bxn[CBX], bmn[CBM];
#pragma unroll
  for (int cb = 0; cb < CBX; ++cb) bx[cb] = ldbuf_bf16x8(rs_wx, lane16, wx0 + (unsigned)cb * KS * 1024u);
#pragma unroll
  for (int cb = 0; cb < CBM; ++cb) bm[cb] = ldbuf_bf16x8(rs_wm, lane16, wm0 + (unsigned)cb * KS * 1024u);
  __syncthreads();

  auto chunk = [&](const int c, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    const char* curx = s_a1 + (size_t)(c & 1) * kA1;
    const char* curm = s_a1 + (size_t)(2 + (c & 1)) * kA1;
    char* nslot = slot_base + (size_t)((c + 1) & 1) * kA1;
    const unsigned kb = (unsigned)(c + 1) * kKC * 4u;   // byte offset of the next chunk's columns
    Unit u;
    if (MORE) unit_load(u, rs_tab, vdst, vsrc, offPx + kb, offQx + kb);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ks = c * 4 + s;
      const unsigned ksn = (unsigned)((!MORE && s == 3) ? ks : ks + 1) * 1024u;
#pragma unroll
      for (int cb = 0; cb < CBX; ++cb) bxn[cb] = ldbuf_bf16x8(rs_wx, lane16, wx0 + (unsigned)cb * KS * 1024u + ksn);
#pragma unroll
      for (int cb = 0; cb < CBM; ++cb) bmn[cb] = ldbuf_bf16x8(rs_wm, lane16, wm0 + (unsigned)cb * KS * 1024u + ksn);
      bf16x8 a[kRB];
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb)
        a[rb] = *reinterpret_cast<const bf16x8*>(curx + ((size_t)(s * 2 + hh) * kRPAD + 32 * rb + r) * 16);
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBX; ++cb)
          accx[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rb], bx[cb], accx[rb][cb], 0, 0, 0);
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb)
        a[rb] = *reinterpret_cast<const bf16x8*>(curm + ((size_t)(s * 2 + hh) * kRPAD + 32 * rb + r) * 16);
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBM; ++cb)
          accm[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rb], bm[cb], accm[rb][cb], 0, 0, 0);
      if (MORE && s == 1) {
        unit_finish(u, s_wdx + (c + 1) * kKC + kg * 8, d2r, nslot);
        unit_load(u, rs_tab, vdst, vsrc, offPm + kb, offQm + kb);
      }
      if (MORE && s == 3) unit_finish(u, s_wdm + (c + 1) * kKC + kg * 8, d2r, nslot + 2 * kA1);
#pragma unroll
      for (int cb = 0; cb < CBX; ++cb) bx[cb] = bxn[cb];
#pragma unroll
      for (int cb = 0; cb < CBM; ++cb) bm[cb] = bmn[cb];
    }
    __syncthreads();
  };
  if (p.dbg & 32) {  // timing experiment: prologue only
    if (s_d2[tid & 63] == 1.2345e-30f) p.agg_x[tid] = 1.f;
    return;
  }
  if (p.dbg & 8) {  // timing experiment: K-loop without the activation build
    for (int c = 0; c < NC; ++c) chunk(c, std::false_type{});
  } else {
    for (int c = 0; c < NC - 1; ++c) chunk(c, std::true_type{});
    chunk(NC - 1, std::false_type{});
  }
  if (p.dbg & 4) {  // timing experiment: stop after the K-loop (keep the accumulators alive)
    float keep = 0.f;
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb) {
#pragma unroll
      for (int cb = 0; cb < CBX; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) keep += accx[rb][cb][i];
#pragma unroll
      for (int cb = 0; cb < CBM; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) keep += accm[rb][cb][i];
    }
    if (keep == 1.2345e-30f) p.agg_x[tid] = keep;
    return;
  }

  // ---- segment bookkeeping (one thread per segment): where does each segment's sum go? ----
  const int S = s_misc[0];
  if (tid < S) {
    const int n = s_seg_node[tid];
    const bool first = (e0 + s_seg_rs[tid]) == p.row_ptr[n];
    const bool last = (e0 + s_seg_re[tid] + 1) == p.row_ptr[n + 1];
    s_seg_mode[tid] = (first && last) ? 2 : (first ? 1 : 0);
  }
  // ---- mlp_x epilogue: s[row] = b3 + sum_n w3[n] * SiLU(acc + b2[n]) ----
  {
    float part[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) part[q] = 0.f;
#pragma unroll
    for (int cb = 0; cb < CBX; ++cb) {
      const int n = 32 * (wave * CBX + cb) + r;
      const float b = p.b2x[n], w = p.w3x[n];
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) part[rb * 16 + i] = fmaf(w, silu_s(fmaf(accx[rb][cb][i], kNegLog2e, b)), part[rb * 16 + i]);
    }
    const float tot = butterfly32(part, lane);   // row-sum of value index q = r over this wave's columns
    s_part[wave * kR + 32 * (r >> 4) + acc_row(r & 15, lane)] = tot;
  }
  // ---- mlp_m epilogue: m = SiLU(acc + b2), gate = sigmoid(wa . m + ba) (:31-34, :59-60) ----
  static_assert(CBM == 1, "in-register message epilogue assumes one 32-column block per wave");
  float mval[32];
  const int ncol = 32 * wave + r;   // message column of this lane
  {
    const float b = p.b2m[ncol], wa = p.wa[ncol];
    float g[32];
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        mval[rb * 16 + i] = silu_s(fmaf(accm[rb][0][i], kNegLog2e, b));   // = -log2(e) * m
        g[rb * 16 + i] = wa * mval[rb * 16 + i];
      }
    const float tot = butterfly32(g, lane);
    s_partg[wave * kR + 32 * (r >> 4) + acc_row(r & 15, lane)] = tot;
  }
  __syncthreads();
  if (tid < kR) {
    float v = p.scal[0], gsum = p.scal[1];
#pragma unroll
    for (int w = 0; w < 8; ++w) { v += s_part[w * kR + tid]; gsum += s_partg[w * kR + tid]; }
    s_sval[tid] = v;
    s_gate[tid] = sigmoid_f(gsum) * kNegInvLog2e;   // also undoes the scale of mval
  }
  __syncthreads();

  // ---- segment sums per receiving node (aggr='sum' into edge_index[0], :10-11) ----
  // A segment holding all edges of its node is stored directly; otherwise it goes to the tile's partial
  // slot (1 = holds the node's first edge, 0 = continues a node begun in an earlier tile) and node_post
  // adds the partials in tile order: fixed summation order, no atomics.
  if (S <= kMaxSegFast) {
    {  // gate weight of every row for every segment (0 outside the segment and for padding rows)
      const int seg = tid >> 6, row = tid & 63;
      if (seg < S) s_gseg[seg * kR + row] = (s_seg_of_row[row] == seg) ? s_gate[row] : 0.f;
    }
    __syncthreads();
    for (int seg = 0; seg < S; ++seg) {
      const float* gw = s_gseg + seg * kR + 4 * hh;
      float v = 0.f;
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) v = fmaf(mval[rb * 16 + i], gw[32 * rb + (i & 3) + 8 * (i >> 2)], v);
      v += __shfl_xor(v, 32);
      if (hh == 0) {
        const int mode = s_seg_mode[seg];
        float* dstp = mode == 2 ? p.agg_m + (size_t)s_seg_node[seg] * p.MP : p.part_m + ((size_t)tile * 2 + mode) * p.MP;
        dstp[ncol] = v;
      }
    }
    if (wave == 7) {  // coordinate messages (x_i - x_j) * s_ij; 1/(G+1) is applied in node_post
      const int row = lane, myseg = s_seg_of_row[row];
      const float sv = s_sval[row];
      const float c0 = s_diff[row] * sv, c1 = s_diff[kR + row] * sv, c2 = s_diff[2 * kR + row] * sv;
      for (int seg = 0; seg < S; ++seg) {
        const bool in = myseg == seg;
        float a0 = in ? c0 : 0.f, a1 = in ? c1 : 0.f, a2 = in ? c2 : 0.f;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { a0 += __shfl_xor(a0, m); a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); }
        if (lane < 3) {
          const int mode = s_seg_mode[seg];
          float* dstp = mode == 2 ? p.agg_x + (size_t)s_seg_node[seg] * 4 : p.part_x + ((size_t)tile * 2 + mode) * 4;
          dstp[lane] = lane == 0 ? a0 : (lane == 1 ? a1 : a2);
        }
      }
    }
  } else {
    // many short segments (sparse graphs): stage the gated messages in LDS, one thread per column
    const int ld = p.MP + 1;
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = 32 * rb + acc_row(i, lane);
        s_msg[row * ld + ncol] = mval[rb * 16 + i] * s_gate[row];
      }
    __syncthreads();
    for (int c = tid; c < p.MP + 3; c += kT2) {
      const bool isx = c >= p.MP;   // three extra "columns" carry the coordinate messages
      const int d = c - p.MP;
      for (int seg = 0; seg < S; ++seg) {
        const int rs = s_seg_rs[seg], re = s_seg_re[seg], mode = s_seg_mode[seg];
        float sum = 0.f;
        for (int rr = rs; rr <= re; ++rr) sum += isx ? s_diff[d * kR + rr] * s_sval[rr] : s_msg[rr * ld + c];
        if (isx) {
          float* dstp = mode == 2 ? p.agg_x + (size_t)s_seg_node[seg] * 4 : p.part_x + ((size_t)tile * 2 + mode) * 4;
          dstp[d] = sum;
        } else {
          float* dstp = mode == 2 ? p.agg_m + (size_t)s_seg_node[seg] * p.MP : p.part_m + ((size_t)tile * 2 + mode) * p.MP;
          dstp[c] = sum;
        }
      }
    }
  }
}

template <int CBX, int CBM>
int launch_v2(const EdgeParams& p, int tiles, hipStream_t st) {
  hipLaunchKernelGGL((edge_kernel_bf16_v2<CBX, CBM>), dim3(tiles), dim3(kT2), v2_smem_bytes(p.MP), st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace

int init_edge_bf16_v2_attributes() {
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v2<1, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v2<2, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v2<4, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return EGNN_OK;
}

bool edge_bf16_v2_supported(const EdgeParams& p) {
  const int cbx = p.WxP / 256;
  return p.WxP == p.WmP && p.WxP % 256 == 0 && (cbx == 1 || cbx == 2 || cbx == 4) && p.MP == 256 &&
         v2_smem_bytes(p.MP) <= 160 * 1024 && (size_t)p.N * p.TC * 4 < ((size_t)1 << 32);
}

int launch_edge_bf16_v2(const EdgeParams& p, int tiles, hipStream_t st) {
  switch (p.WxP / 256) {
    case 1: return launch_v2<1, 1>(p, tiles, st);
    case 2: return launch_v2<2, 1>(p, tiles, st);
    case 4: return launch_v2<4, 1>(p, tiles, st);
  }
  set_error("edge_kernel_bf16_v2: unsupported width %d", p.WxP);
  return EGNN_EINVAL;
}

}  // namespace egnn
