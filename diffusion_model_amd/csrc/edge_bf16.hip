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
#include "kernels.h"

namespace egnn {

namespace {

constexpr int kT2 = 512;  // threads
constexpr int kR = 64, kRB = 2, kRPAD = kR + 1, kKC = 64;
constexpr size_t kA1 = (size_t)8 * kRPAD * 16;  // bytes of one activation chunk [8 k-groups][65][8 bf16]

__host__ __device__ inline size_t v2_small_bytes() { return (size_t)(2 + 1 + 3 + 1 + 1 + 8) * kR * 4; }
__host__ __device__ inline size_t v2_smem_bytes(int MP) { return v2_small_bytes() + 4 * kA1 + (size_t)kR * (MP + 1) * 4; }

struct Unit {  // one build unit in flight: 8 columns of one row of one MLP
  f32x4 p0, p1, q0, q1, w0, w1;
};

__device__ __forceinline__ void unit_load(Unit& u, const float* __restrict__ prow, const float* __restrict__ qrow,
                                          const float* __restrict__ wd, int k0) {
  u.p0 = *reinterpret_cast<const f32x4*>(prow + k0);
  u.p1 = *reinterpret_cast<const f32x4*>(prow + k0 + 4);
  u.q0 = *reinterpret_cast<const f32x4*>(qrow + k0);
  u.q1 = *reinterpret_cast<const f32x4*>(qrow + k0 + 4);
  u.w0 = *reinterpret_cast<const f32x4*>(wd + k0);
  u.w1 = *reinterpret_cast<const f32x4*>(wd + k0 + 4);
}
__device__ __forceinline__ void unit_finish(const Unit& u, float d2, char* slot) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[j] = (__bf16)silu_f(fmaf(u.w0[j], d2, u.p0[j] + u.q0[j]));
    o[j + 4] = (__bf16)silu_f(fmaf(u.w1[j], d2, u.p1[j] + u.q1[j]));
  }
  *reinterpret_cast<bf16x8*>(slot) = o;
}

template <int CBX, int CBM>
__global__ __launch_bounds__(kT2, 2) void edge_kernel_bf16_v2(const EdgeParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_dst = reinterpret_cast<int*>(smem);
  int* s_src = s_dst + kR;
  float* s_d2 = reinterpret_cast<float*>(s_src + kR);
  float* s_diff = s_d2 + kR;  // [3][R]
  float* s_sval = s_diff + 3 * kR;
  float* s_gate = s_sval + kR;
  float* s_part = s_gate + kR;  // [8][R]
  char* s_a1 = smem + v2_small_bytes();  // [x0, x1, m0, m1]
  float* s_msg = reinterpret_cast<float*>(s_a1 + 4 * kA1);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  }
  __syncthreads();

  // ---- fused K-loop ----
  const int KP = p.WxP;            // == p.WmP (checked on the host)
  const int NC = KP / kKC, KS = KP / 16;
  const int brow = tid >> 3, kg = tid & 7;   // build unit of this thread
  const float* trow_d = p.table + (size_t)s_dst[brow] * p.TC;
  const float* trow_s = p.table + (size_t)s_src[brow] * p.TC;
  const float d2r = s_d2[brow];
  const int offPx = 0, offQx = p.WxP, offPm = 2 * p.WxP, offQm = 2 * p.WxP + p.WmP;
  char* slot_base = s_a1 + ((size_t)kg * kRPAD + brow) * 16;  // + buffer offset

  const bf16x8* wbx[CBX];
  const bf16x8* wbm[CBM];
  {
    const bf16x8* w2x = reinterpret_cast<const bf16x8*>(p.w2x);
    const bf16x8* w2m = reinterpret_cast<const bf16x8*>(p.w2m);
#pragma unroll
    for (int cb = 0; cb < CBX; ++cb) wbx[cb] = w2x + ((size_t)(wave * CBX + cb) * KS) * 64 + lane;
#pragma unroll
    for (int cb = 0; cb < CBM; ++cb) wbm[cb] = w2m + ((size_t)(wave * CBM + cb) * KS) * 64 + lane;
  }

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
    unit_load(u, trow_d + offPx, trow_s + offQx, p.wdx, kg * 8);
    unit_finish(u, d2r, slot_base);
    unit_load(u, trow_d + offPm, trow_s + offQm, p.wdm, kg * 8);
    unit_finish(u, d2r, slot_base + 2 * kA1);
  }
  bf16x8 bx[CBX], bm[CBM], bxn[CBX], bmn[CBM];
#pragma unroll
  for (int cb = 0; cb < CBX; ++cb) bx[cb] = wbx[cb][0];
#pragma unroll
  for (int cb = 0; cb < CBM; ++cb) bm[cb] = wbm[cb][0];
  __syncthreads();

  for (int c = 0; c < NC; ++c) {
    const char* curx = s_a1 + (size_t)(c & 1) * kA1;
    const char* curm = s_a1 + (size_t)(2 + (c & 1)) * kA1;
    char* nslot = slot_base + (size_t)((c + 1) & 1) * kA1;
    const bool more = c + 1 < NC;
    const int k0n = (c + 1) * kKC + kg * 8;
    Unit u;
    if (more) unit_load(u, trow_d + offPx, trow_s + offQx, p.wdx, k0n);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ks = c * 4 + s;
      const int ksn = ks + 1 < KS ? ks + 1 : ks;
#pragma unroll
      for (int cb = 0; cb < CBX; ++cb) bxn[cb] = wbx[cb][(size_t)ksn * 64];
#pragma unroll
      for (int cb = 0; cb < CBM; ++cb) bmn[cb] = wbm[cb][(size_t)ksn * 64];
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
      if (s == 1 && more) {
        unit_finish(u, d2r, nslot);
        unit_load(u, trow_d + offPm, trow_s + offQm, p.wdm, k0n);
      }
      if (s == 3 && more) unit_finish(u, d2r, nslot + 2 * kA1);
#pragma unroll
      for (int cb = 0; cb < CBX; ++cb) bx[cb] = bxn[cb];
#pragma unroll
      for (int cb = 0; cb < CBM; ++cb) bm[cb] = bmn[cb];
    }
    __syncthreads();
  }

  // ---- mlp_x epilogue: s[row] = b3 + sum_n w3[n] * SiLU(acc + b2[n]) ----
  {
    float part[kRB][16];
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) part[rb][i] = 0.f;
#pragma unroll
    for (int cb = 0; cb < CBX; ++cb) {
      const int n = 32 * (wave * CBX + cb) + r;
      const float b = p.b2x[n], w = p.w3x[n];
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) part[rb][i] = fmaf(w, silu_f(accx[rb][cb][i] + b), part[rb][i]);
    }
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = part[rb][i];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
        if (r == 0) s_part[wave * kR + 32 * rb + acc_row(i, lane)] = v;
      }
  }
  // ---- mlp_m epilogue: m = SiLU(acc + b2) into the LDS message tile ----
  {
    const int ld = p.MP + 1;
#pragma unroll
    for (int cb = 0; cb < CBM; ++cb) {
      const int n = 32 * (wave * CBM + cb) + r;
      const float b = p.b2m[n];
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s_msg[(32 * rb + acc_row(i, lane)) * ld + n] = silu_f(accm[rb][cb][i] + b);
    }
  }
  __syncthreads();
  if (tid < kR) {
    float v = p.scal[0];
#pragma unroll
    for (int w = 0; w < 8; ++w) v += s_part[w * kR + tid];
    s_sval[tid] = v;
  }
  {  // attention gate, 8 threads per row
    const int row = tid >> 3, sub = tid & 7, ld = p.MP + 1;
    float s = 0.f;
    for (int c = sub; c < p.MP; c += 8) s = fmaf(p.wa[c], s_msg[row * ld + c], s);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (sub == 0) s_gate[row] = sigmoid_f(s + p.scal[1]);
  }
  __syncthreads();

  // ---- segment sums per receiving node (same rule as the fp32 kernel) ----
  auto flush = [&](int n, int rs, int re, float v, float* agg, float* part, int ld, int c) {
    const bool first = (e0 + rs) == p.row_ptr[n];
    const bool last = (e0 + re + 1) == p.row_ptr[n + 1];
    if (first && last) agg[(size_t)n * ld + c] = v;
    else part[((size_t)tile * 2 + (first ? 1 : 0)) * ld + c] = v;
  };
  const int ld = p.MP + 1;
  for (int c = tid; c < p.MP; c += kT2) {
    float sum = 0.f;
    int rs = 0;
    for (int rr = 0; rr < nvalid; ++rr) {
      sum = fmaf(s_msg[rr * ld + c], s_gate[rr], sum);
      if (rr == nvalid - 1 || s_dst[rr + 1] != s_dst[rr]) {
        flush(s_dst[rr], rs, rr, sum, p.agg_m, p.part_m, p.MP, c);
        sum = 0.f;
        rs = rr + 1;
      }
    }
  }
  if (tid >= 448 && tid < 451) {  // coordinate messages on an otherwise idle wave
    const int d = tid - 448;
    float sum = 0.f;
    int rs = 0;
    for (int rr = 0; rr < nvalid; ++rr) {
      sum = fmaf(s_diff[d * kR + rr], s_sval[rr], sum);
      if (rr == nvalid - 1 || s_dst[rr + 1] != s_dst[rr]) {
        flush(s_dst[rr], rs, rr, sum, p.agg_x, p.part_x, 4, d);
        sum = 0.f;
        rs = rr + 1;
      }
    }
  }
}

template <int CBX, int CBM>
int launch_v2(const EdgeParams& p, int tiles, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_kernel_bf16_v2<CBX, CBM>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL((edge_kernel_bf16_v2<CBX, CBM>), dim3(tiles), dim3(kT2), v2_smem_bytes(p.MP), st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace

bool edge_bf16_v2_supported(const EdgeParams& p) {
  const int cbx = p.WxP / 256;
  return p.WxP == p.WmP && p.WxP % 256 == 0 && (cbx == 1 || cbx == 2 || cbx == 4) && p.MP == 256 &&
         v2_smem_bytes(p.MP) <= 160 * 1024;
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
