// Bookkeeping shared by the 128-edge-tile bf16 edge kernels (gfx950): the LDS carve of the per-tile arrays, the tile
// prologue (edge rows, geometry of EquivariantGraphNeuralNetwork.py:56, segment = receiving-node structure of the CSR
// tile) and the coordinate branch's segment sums (:62-65, aggregate 'sum' of :70).  One implementation for every kernel
// that works on such a tile; the K loops and the accumulator-layout-specific row reductions stay in the kernels.
#pragma once
#include "kernels.h"

namespace egnn {
namespace tile128 {

constexpr int kR = 128;         // edges per tile
constexpr int kSegFast = 8;     // segments (receiving nodes) per tile handled by the register path
// LDS carve (bytes) of the per-tile arrays; the kernel's own K-loop buffers start at kOffLoop
constexpr size_t kOffDst = 0;                                  // int[R]
constexpr size_t kOffSrc = kOffDst + kR * 4;                   // int[R]
constexpr size_t kOffD2 = kOffSrc + kR * 4;                    // float[R]
constexpr size_t kOffDiff = kOffD2 + kR * 4;                   // float[3][R]
constexpr size_t kOffVal = kOffDiff + 3 * kR * 4;              // float[R]   s_ij (X) / gate (M)
constexpr size_t kOffPart = kOffVal + kR * 4;                  // float[8][R] per-wave partial row sums
constexpr size_t kOffSegRow = kOffPart + 8 * kR * 4;           // int[R]
constexpr size_t kOffSegNode = kOffSegRow + kR * 4;            // int[R]
constexpr size_t kOffSegRs = kOffSegNode + kR * 4;             // int[R]
constexpr size_t kOffSegRe = kOffSegRs + kR * 4;               // int[R]
constexpr size_t kOffSegMode = kOffSegRe + kR * 4;             // int[R]
constexpr size_t kOffMisc = kOffSegMode + kR * 4;              // int[16]
constexpr size_t kOffGseg = kOffMisc + 64;                     // float[kSegFast][R]
constexpr size_t kOffLoop = kOffGseg + kSegFast * kR * 4;

struct Lds {
  int *dst, *src, *seg_of_row, *seg_node, *seg_rs, *seg_re, *seg_mode, *misc;
  float *d2, *diff, *val, *part, *gseg;
  __device__ __forceinline__ explicit Lds(char* smem)
      : dst(reinterpret_cast<int*>(smem + kOffDst)), src(reinterpret_cast<int*>(smem + kOffSrc)),
        seg_of_row(reinterpret_cast<int*>(smem + kOffSegRow)), seg_node(reinterpret_cast<int*>(smem + kOffSegNode)),
        seg_rs(reinterpret_cast<int*>(smem + kOffSegRs)), seg_re(reinterpret_cast<int*>(smem + kOffSegRe)),
        seg_mode(reinterpret_cast<int*>(smem + kOffSegMode)), misc(reinterpret_cast<int*>(smem + kOffMisc)),
        d2(reinterpret_cast<float*>(smem + kOffD2)), diff(reinterpret_cast<float*>(smem + kOffDiff)),
        val(reinterpret_cast<float*>(smem + kOffVal)), part(reinterpret_cast<float*>(smem + kOffPart)),
        gseg(reinterpret_cast<float*>(smem + kOffGseg)) {}
};

// Edge rows [e0, e0 + nvalid) of the tile: indices, x_i - x_j, d2 = norm(x_i - x_j)**2 (sqrt then square, as :56), the
// d^2 column of the first layer (wd, KP entries) into s_wd, and the segment structure: consecutive rows with the same
// receiving node form a segment; seg_mode = 2 when the segment holds ALL edges of its node (sum goes to the node's slot),
// 1 / 0 when the node's edges start here / continue from the previous tile (sum goes to the tile's partial slots, added
// in tile order by node_post: bitwise deterministic, no atomics).  Returns the number of segments.
// 512 threads; ends with every thread past the last barrier EXCEPT for seg_mode (read it after the caller's next barrier).
// Split in two so that a kernel can put its first table / weight requests between the halves (prologue_rows ends behind the
// barrier that publishes L.dst / L.src / L.d2; prologue_segments is two more barriers with only waves 0 and 1 busy).
__device__ __forceinline__ void prologue_rows(const EdgeParams& p, const Lds& L, int e0, int nvalid, const float* wd, int KP,
                                              float* s_wd, int tid) {
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
    L.dst[tid] = d;
    L.src[tid] = s;
    L.diff[tid] = dx; L.diff[kR + tid] = dy; L.diff[2 * kR + tid] = dz;
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    L.d2[tid] = nrm * nrm;
  }
  for (int i = tid; i < KP; i += 512) s_wd[i] = wd[i];
  __syncthreads();
}

// seg_mode of segment `seg` (2 = all edges of its node, 1 = the node's edges start here, 0 = continue from the previous tile):
// two row_ptr loads; prologue_segments<false> leaves it to the caller, who can put the loads under its K loop.
__device__ __forceinline__ int segment_mode(const EdgeParams& p, const Lds& L, int e0, int seg) {
  const int n = L.seg_node[seg];
  const bool first = (e0 + L.seg_rs[seg]) == p.row_ptr[n];
  const bool last = (e0 + L.seg_re[seg] + 1) == p.row_ptr[n + 1];
  return (first && last) ? 2 : (first ? 1 : 0);
}

template <bool WITH_MODES = true>
__device__ __forceinline__ int prologue_segments(const EdgeParams& p, const Lds& L, int e0, int nvalid, int tid, int lane, int wave) {
  bool is_start = false, is_end = false;
  unsigned long long starts = 0;
  if (tid < kR) {   // waves 0 and 1
    const bool valid = tid < nvalid;
    const int d = L.dst[tid];
    is_start = valid && (tid == 0 || L.dst[tid - 1] != d);
    is_end = valid && (tid == nvalid - 1 || L.dst[tid + 1] != d);
    starts = __ballot(is_start);
    if (lane == 0) L.misc[1 + wave] = __popcll(starts);
  }
  __syncthreads();
  if (tid < kR) {
    const int seg = (wave == 1 ? L.misc[1] : 0) + __popcll(starts & ((2ull << lane) - 1ull)) - 1;
    L.seg_of_row[tid] = tid < nvalid ? seg : -1;
    if (is_start) { L.seg_node[seg] = L.dst[tid]; L.seg_rs[seg] = tid; }
    if (is_end) L.seg_re[seg] = tid;
    if (tid == 0) L.misc[0] = L.misc[1] + L.misc[2];
  }
  __syncthreads();
  const int S = L.misc[0];
  if constexpr (WITH_MODES) {
    if (tid < S) L.seg_mode[tid] = segment_mode(p, L, e0, tid);
  }
  return S;
}

__device__ __forceinline__ int prologue(const EdgeParams& p, const Lds& L, int e0, int nvalid, const float* wd, int KP,
                                        float* s_wd, int tid, int lane, int wave) {
  prologue_rows(p, L, e0, nvalid, wd, KP, s_wd, tid);
  return prologue_segments(p, L, e0, nvalid, tid, lane, wave);
}

// Coordinate messages of the tile (:62-65): sum over each segment's rows of (x_i - x_j) * s_ij (L.val) into the node's /
// the tile's slot of copy `half` of the coordinate sums; component 3 carries the segment's sum of |x_i - x_j|^2 (plain
// squares: the Frobenius norm of :64 is the root of the sum over ALL edges), so the normaliser needs no edge pass of its
// own.  1/(G+1) is applied in node_post.  Call after a barrier that orders L.val; 512 threads.
__device__ __forceinline__ void coordinate_segment_sums(const EdgeParams& p, const Lds& L, int S, int tile, int half, int tid,
                                                        int lane, int wave) {
  float* aggx = p.agg_x + (size_t)half * p.agg_x_stride;
  float* partx = p.part_x + (size_t)half * p.part_x_stride;
  if (S <= kSegFast) {
    // one wave per segment (S <= 8 = the waves of the workgroup): the 6-step cross-lane sums of the segments run side by side
    // (one wave doing the segments one after the other was 1.4 us of a 27 us workgroup, the other 7 waves idle:
    // tools/fwd_stamps.py)
    if (wave < S) {
      const int seg = wave;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int row = lane + 64 * u;
        if (L.seg_of_row[row] == seg) {
          const float sv = L.val[row];
          const float dx = L.diff[row], dy = L.diff[kR + row], dz = L.diff[2 * kR + row];
          a0 += dx * sv; a1 += dy * sv; a2 += dz * sv;
          a3 += dx * dx + dy * dy + dz * dz;
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) { a0 += __shfl_xor(a0, m); a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); a3 += __shfl_xor(a3, m); }
      if (lane < 4) {
        const int mode = L.seg_mode[seg];
        float* dstp = mode == 2 ? aggx + (size_t)L.seg_node[seg] * 4 : partx + ((size_t)tile * 2 + mode) * 4;
        dstp[lane] = lane == 0 ? a0 : (lane == 1 ? a1 : (lane == 2 ? a2 : a3));
      }
    }
  } else {
    for (int t = tid; t < 4 * S; t += 512) {
      const int seg = t >> 2, d = t & 3, mode = L.seg_mode[seg];
      float sum = 0.f;
      if (d < 3) {
        for (int rr = L.seg_rs[seg]; rr <= L.seg_re[seg]; ++rr) sum += L.diff[d * kR + rr] * L.val[rr];
      } else {
        for (int rr = L.seg_rs[seg]; rr <= L.seg_re[seg]; ++rr) {
          const float dx = L.diff[rr], dy = L.diff[kR + rr], dz = L.diff[2 * kR + rr];
          sum += dx * dx + dy * dy + dz * dz;
        }
      }
      float* dstp = mode == 2 ? aggx + (size_t)L.seg_node[seg] * 4 : partx + ((size_t)tile * 2 + mode) * 4;
      dstp[d] = sum;
    }
  }
}


// ---- epilogues on 32x32 accumulator tiles (v_mfma_f32_32x32x16_bf16: column = lane & 31, row = acc_row(reg, lane)) ------
// wave `wave` of 8 holds acc[rb][cb] = 4 row blocks x CB column blocks (columns 32 (colblk0 + cb) + (lane & 31)) of
// t2 / -log2(e) (the accumulators of the scaled second-layer product); p carries the SCALED parameter vectors.

// Head of the coordinate branch (:62-63): L.val[row] = [b3 +] sum over this workgroup's columns of w3[n] SiLU(a2[row][n]),
// the bias only in column share 0.  Ends behind a barrier (L.val ordered for every thread).
// acc_scale: t2 = acc_scale * acc + b2 (-log2(e); the fp16 streams of precision f16c8 carry 2^8, divided out here)
template <int CB>
__device__ __forceinline__ void x_head(const EdgeParams& p, const Lds& L, const f32x16 (&acc)[4][CB], int colblk0, int half,
                                       int tid, int lane, int wave, const float acc_scale = kNegLog2e) {
  const int r = lane & 31;
  float part[64];
#pragma unroll
  for (int q = 0; q < 64; ++q) part[q] = 0.f;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const int n = 32 * (colblk0 + cb) + r;
    const float bb = p.b2x[n], w = p.w3x[n];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) part[rb * 16 + i] = fmaf(w, silu_s(fmaf(acc[rb][cb][i], acc_scale, bb)), part[rb * 16 + i]);
  }
  {
    float lo[32], hi[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) { lo[q] = part[q]; hi[q] = part[32 + q]; }
    const float t0 = butterfly32(lo, lane), t1 = butterfly32(hi, lane);
    const int row = 32 * (r >> 4) + acc_row(r & 15, lane);   // row of value index q = lane & 31
    L.part[wave * kR + row] = t0;
    L.part[wave * kR + 64 + row] = t1;
  }
  __syncthreads();
  if (tid < kR) {
    float v = half == 0 ? p.scal[0] : 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += L.part[w * kR + tid];
    L.val[tid] = v;
  }
  __syncthreads();
}

// Message branch (:57-61): m = SiLU(a2 + b2), gate = sigmoid(wa . m + ba), segment sums of m * gate into agg_m / part_m.
// One 32-column block per wave (256 message columns per workgroup).  Segments are summed kSegFast per pass: the gate of
// each row is laid out per segment in LDS (L.gseg) and every lane dots its 64 message values with it -- no staging tile.
// PRE_SCALED: acc already holds t2 = -log2(e) * (a2 + b2) (the training forward stores it before this call).
// acc_scale: t2 = acc_scale * acc + b2 (-log2(e); precision fp16 divides the weight fragments' 2^8 out here).
template <bool PRE_SCALED = false>
__device__ __forceinline__ void message_epilogue(const EdgeParams& p, const Lds& L, const f32x16 (&acc)[4][1], int S, int tile,
                                                 int tid, int lane, int wave, const float acc_scale = kNegLog2e,
                                                 const int colblk = -1) {
  const int r = lane & 31, hh = lane >> 5;
  const int ncol = 32 * (colblk < 0 ? wave : colblk) + r;   // colblk: the wave's column block when it is not its index
  float mval[64];
  {
    const float bb = p.b2m[ncol], wa = p.wa[ncol];
    float lo[32], hi[32];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float m = silu_s(PRE_SCALED ? acc[rb][0][i] : fmaf(acc[rb][0][i], acc_scale, bb));   // = -log2(e) * m
        mval[rb * 16 + i] = m;
        if (rb < 2) lo[rb * 16 + i] = wa * m; else hi[(rb - 2) * 16 + i] = wa * m;
      }
    const float t0 = butterfly32(lo, lane), t1 = butterfly32(hi, lane);
    const int row = 32 * (r >> 4) + acc_row(r & 15, lane);
    L.part[wave * kR + row] = t0;
    L.part[wave * kR + 64 + row] = t1;
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
    for (int t = tid; t < ns * kR; t += 512) {
      const int seg = base + (t >> 7), row = t & 127;
      L.gseg[t] = (L.seg_of_row[row] == seg) ? L.val[row] : 0.f;
    }
    __syncthreads();
    for (int sg = 0; sg < ns; ++sg) {
      const int seg = base + sg;
      const float* gw = L.gseg + sg * kR + 4 * hh;
      float v = 0.f;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) v = fmaf(mval[rb * 16 + i], gw[32 * rb + (i & 3) + 8 * (i >> 2)], v);
      v += __shfl_xor(v, 32);
      if (hh == 0) {
        const int mode = L.seg_mode[seg];
        float* dstp = mode == 2 ? p.agg_m + (size_t)L.seg_node[seg] * p.MP : p.part_m + ((size_t)tile * 2 + mode) * p.MP;
        dstp[ncol] = v;
      }
    }
  }
}

// sum over the 16 lanes of a DPP row of value index j = 2 * (lane & 15) + {0, 1} of v[32]: halving butterfly on row_ror:8,
// row_half_mirror and quad_perm moves with bank-masked selects (the lower four stages of butterfly32), no LDS.
__device__ __forceinline__ void butterfly16(float (&v)[32], int lane, float& out0, float& out1) {
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float ta = v[q] + dpp_mov<0x128>(v[q]);          // row_ror:8  (lane ^ 8)
    const float tb = v[q + 16] + dpp_mov<0x128>(v[q + 16]);
    v[q] = dpp_sel<0xC>(ta, tb);                            // lanes with (lane & 8) keep index q + 16
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float ta = v[q] + dpp_mov<0x141>(v[q]);          // row_half_mirror (pairs lane l with 7 - l)
    const float tb = v[q + 8] + dpp_mov<0x141>(v[q + 8]);
    v[q] = dpp_sel<0xA>(ta, tb);                            // lanes with (lane & 4) keep index q + 8
  }
  const bool b2 = (lane & 2) != 0, b1 = (lane & 1) != 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float ta = v[q] + dpp_mov<0x4E>(v[q]);           // quad_perm [2,3,0,1]
    const float tb = v[q + 4] + dpp_mov<0x4E>(v[q + 4]);
    v[q] = b2 ? tb : ta;                                    // lanes with (lane & 2) keep index q + 4
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float ta = v[q] + dpp_mov<0xB1>(v[q]);           // quad_perm [1,0,3,2]
    const float tb = v[q + 2] + dpp_mov<0xB1>(v[q + 2]);
    v[q] = b1 ? tb : ta;                                    // lanes with (lane & 1) keep index q + 2
  }
  out0 = v[0];
  out1 = v[1];
}

}  // namespace tile128
}  // namespace egnn
