// Per-edge kernels at fp32-grade accuracy on the bf16 matrix cores (gfx950): precision 'bf16x3'.
//
// north_star asks for eps within 1e-4 of the reference's fp32 path.  The fp32 kernels meet it on v_mfma_f32_32x32x2_f32,
// which runs at 1/16 of the bf16 MFMA rate (109 ms per C2 step against 10.5 ms for plain bf16, whose operands carry 8
// significant bits: 2-4e-3 on eps).  Here both operands of the two per-edge second-layer products
// (EquivariantGraphNeuralNetwork.py:15-16 mlp_m.2, :21-22 mlp_x.2) are split into a bf16 head and a bf16 remainder,
//     a = a_hi + a_lo,   W = W_hi + W_lo,      a . W  ~=  a_lo . W_hi + a_hi . W_lo + a_hi . W_hi        (fp32 accumulate)
// -- three v_mfma_f32_32x32x16_bf16 per tile instead of one; the dropped a_lo . W_lo term and the remainders' own rounding
// are 2^-17 relative.  Everything else is the fp32 path's arithmetic: fp32 first-layer table (exact f32 MFMA in node_pre),
// fp32 geometry, SiLU / sigmoid and segment sums, fp32 node MLP.  The tile, the phase-opposed K loop and the operand
// pipeline are those of edge_bf16_v3.hip at ONE 32-column block per wave (256 columns per workgroup: the coordinate
// branch runs as WxP / 256 column shares whose coordinate sums node_post adds; s_ij is linear in the column shares);
// prologue, heads and segment sums are the shared ones of edge_tile.h.
#include "edge_tile.h"

namespace egnn {

namespace {

using namespace tile128;
constexpr int kT = 512;
constexpr int kRPAD = kR + 1;
constexpr int kKC = 64;
constexpr size_t kImg = (size_t)8 * kRPAD * 16;   // one activation image [8 k-groups][129][8 bf16] (head or remainder)
__host__ __device__ inline size_t x3_smem_bytes(int KP) { return kOffLoop + 4 * kImg + (size_t)KP * 4; }

// SiLU + head / remainder split of one build unit (8 columns of one row): table, wd pre-scaled by -log2(e)
__device__ __forceinline__ void unit_finish_hilo(const Unit& u, const float* wd, float d2, char* slot_hi, char* slot_lo) {
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(wd), w1 = *reinterpret_cast<const f32x4*>(wd + 4);
  bf16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = silu_s(fmaf(w0[j], d2, u.p0[j] + u.q0[j]));
    const float b = silu_s(fmaf(w1[j], d2, u.p1[j] + u.q1[j]));
    hi[j] = (__bf16)a;
    hi[j + 4] = (__bf16)b;
    lo[j] = (__bf16)(a - (float)hi[j]);
    lo[j + 4] = (__bf16)(b - (float)hi[j + 4]);
  }
  *reinterpret_cast<bf16x8*>(slot_hi) = hi;
  *reinterpret_cast<bf16x8*>(slot_lo) = lo;
}

template <bool IS_M>
__global__ __launch_bounds__(kT, 2) void edge_x3_kernel(const EdgeParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Lds L(smem);
  char* s_a1 = smem + kOffLoop;                       // [2 buffers][head image | remainder image]
  float* s_wd = reinterpret_cast<float*>(s_a1 + 4 * kImg);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int KP = IS_M ? p.WmP : p.WxP;
  const int nsplit = IS_M ? 1 : p.WxP / 256;
  const int j = xcd_tile(blockIdx.x, gridDim.x);
  const int tile = j / nsplit, half = j - tile * nsplit;
  const int e0 = tile * kR;
  const int nvalid = min(kR, p.E - e0);

  const int S = prologue(p, L, e0, nvalid, IS_M ? p.wdm : p.wdx, KP, s_wd, tid, lane, wave);

  // ---- K loop ----
  const int NC = KP / kKC, KS = KP / 16;
  const int brow = tid >> 3, kg = tid & 7;   // this thread builds rows brow and brow + 64, columns 8 kg .. 8 kg + 7 of a chunk
  const rsrc_t rs_tab = make_rsrc(p.table, (unsigned)min((size_t)p.N * p.TC * 4, (size_t)0xFFFFFFFFu));
  const size_t wbytes = (size_t)(IS_M ? p.MP : p.WxP) * KP * 2;
  const rsrc_t rs_wh = make_rsrc(IS_M ? p.w2m : p.w2x, (unsigned)wbytes);
  const rsrc_t rs_wl = make_rsrc(IS_M ? p.w2m_lo : p.w2x_lo, (unsigned)wbytes);
  const unsigned vdst0 = (unsigned)L.dst[brow] * (unsigned)p.TC * 4u + (unsigned)kg * 32u;
  const unsigned vsrc0 = (unsigned)L.src[brow] * (unsigned)p.TC * 4u + (unsigned)kg * 32u;
  const unsigned vdst1 = (unsigned)L.dst[brow + 64] * (unsigned)p.TC * 4u + (unsigned)kg * 32u;
  const unsigned vsrc1 = (unsigned)L.src[brow + 64] * (unsigned)p.TC * 4u + (unsigned)kg * 32u;
  const float d2r0 = L.d2[brow], d2r1 = L.d2[brow + 64];
  const unsigned offP = (IS_M ? 2u * p.WxP : 0u) * 4u, offQ = (IS_M ? 2u * p.WxP + p.WmP : (unsigned)p.WxP) * 4u;   // fp32 table
  char* slot0 = s_a1 + ((size_t)kg * kRPAD + brow) * 16;
  char* slot1 = slot0 + 64 * 16;
  const unsigned lane16 = lane * 16u;
  const unsigned lds_a1 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(s_a1 + ((size_t)hh * kRPAD + r) * 16);
  const int colblk0 = half * 8 + wave;   // this wave's 32-column block
  const unsigned w0 = (unsigned)colblk0 * KS * 1024u;

  f32x16 acc[4][1];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[rb][0][i] = 0.f;

  {  // chunk 0
    Unit u;
    unit_load(u, rs_tab, vdst0, vsrc0, offP, offQ);
    unit_finish_hilo(u, s_wd + kg * 8, d2r0, slot0, slot0 + kImg);
    unit_load(u, rs_tab, vdst1, vsrc1, offP, offQ);
    unit_finish_hilo(u, s_wd + kg * 8, d2r1, slot1, slot1 + kImg);
  }
  bf16x8 bh[4], bl[4];   // head / remainder weight fragments of the 4 k-steps of the current chunk
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bh[s] = ldbuf_bf16x8(rs_wh, lane16, w0 + (unsigned)s * 1024u);
    bl[s] = ldbuf_bf16x8(rs_wl, lane16, w0 + (unsigned)s * 1024u);
  }
  __syncthreads();

  // matrix phase of chunk c: 4 k-steps x 4 row blocks x 3 products.  Operand pipeline as in edge_bf16_v3.hip: the (head,
  // remainder) operand pair of a row block by inline-asm ds_read_b128, refilled in place for the next k-step right after
  // the row block's MFMAs were issued; LDS returns in order: lgkmcnt(6) = all but the 3 younger pairs have landed.
  auto mphase = [&](const int c, const bool last) {
    const unsigned abase = lds_a1 + (unsigned)(c & 1) * (unsigned)(2 * kImg);
    bf16x8 ah[4], al[4];
#define LDS_RD(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(abase), "n"(off))
#define LDS_RD2(rb, off) LDS_RD(ah[rb], off); LDS_RD(al[rb], (off) + (int)kImg)
#define LDS_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
    LDS_RD2(0, 0); LDS_RD2(1, 512); LDS_RD2(2, 1024); LDS_RD2(3, 1536);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        if (s < 3 || rb == 0) LDS_WAIT(6);
        else if (rb == 1) LDS_WAIT(4);
        else if (rb == 2) LDS_WAIT(2);
        else LDS_WAIT(0);
        asm volatile("" : "+v"(ah[rb]), "+v"(al[rb]));   // uses stay below the wait
        acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rb], bh[s], acc[rb][0], 0, 0, 0);
        acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rb], bl[s], acc[rb][0], 0, 0, 0);
        acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rb], bh[s], acc[rb][0], 0, 0, 0);
        // k-step s + 1 of the same chunk: k-groups 2 (s + 1), 2 (s + 1) + 1 -> + 2 * 129 * 16 bytes per k-step
        if (s == 0) { if (rb == 0) { LDS_RD2(0, 4128); } if (rb == 1) { LDS_RD2(1, 4128 + 512); } if (rb == 2) { LDS_RD2(2, 4128 + 1024); } if (rb == 3) { LDS_RD2(3, 4128 + 1536); } }
        if (s == 1) { if (rb == 0) { LDS_RD2(0, 8256); } if (rb == 1) { LDS_RD2(1, 8256 + 512); } if (rb == 2) { LDS_RD2(2, 8256 + 1024); } if (rb == 3) { LDS_RD2(3, 8256 + 1536); } }
        if (s == 2) { if (rb == 0) { LDS_RD2(0, 12384); } if (rb == 1) { LDS_RD2(1, 12384 + 512); } if (rb == 2) { LDS_RD2(2, 12384 + 1024); } if (rb == 3) { LDS_RD2(3, 12384 + 1536); } }
      }
      if (!last) {
        const unsigned ksn = (unsigned)((c + 1) * 4 + s) * 1024u;
        bh[s] = ldbuf_bf16x8(rs_wh, lane16, w0 + ksn);
        bl[s] = ldbuf_bf16x8(rs_wl, lane16, w0 + ksn);
      }
    }
#undef LDS_WAIT
#undef LDS_RD2
#undef LDS_RD
  };
  Unit ua0, ua1;
  auto vload = [&](const int cq) {   // fp32 table rows for the activations of chunk cq (clamped: a harmless repeat at the end)
    const int c = cq < NC ? cq : NC - 1;
    const unsigned kb = (unsigned)c * kKC * 4u;
    unit_load(ua0, rs_tab, vdst0, vsrc0, offP + kb, offQ + kb);
    unit_load(ua1, rs_tab, vdst1, vsrc1, offP + kb, offQ + kb);
  };
  auto vfinish = [&](const int c) {
    const size_t nbuf = (size_t)(c & 1) * (2 * kImg);
    __builtin_amdgcn_s_setprio(3);
    unit_finish_hilo(ua0, s_wd + c * kKC + kg * 8, d2r0, slot0 + nbuf, slot0 + nbuf + kImg);
    unit_finish_hilo(ua1, s_wd + c * kKC + kg * 8, d2r1, slot1 + nbuf, slot1 + nbuf + kImg);
    __builtin_amdgcn_s_setprio(0);
  };
  // SIMD partners (waves w and w + 4) in opposite phase, one barrier per chunk (edge_bf16_v3.hip)
  vload(1);
  if (wave < 4) {
    for (int i = 0; i < NC - 1; ++i) {
      mphase(i, false);
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

  if constexpr (IS_M) {
    message_epilogue(p, L, acc, S, tile, tid, lane, wave);
  } else {
    x_head<1>(p, L, acc, colblk0, half, tid, lane, wave);
    coordinate_segment_sums(p, L, S, tile, half, tid, lane, wave);
  }
}

}  // namespace

int init_edge_bf16x3_attributes() {
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
  EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
  return EGNN_OK;
}

bool edge_bf16x3_supported(const EdgeParams& p) {
  return (p.WxP == 256 || p.WxP == 512 || p.WxP == 1024) && p.MP == 256 && p.WmP % 64 == 0 && p.w2x_lo && p.w2m_lo &&
         x3_smem_bytes(p.WxP) <= 160 * 1024 && x3_smem_bytes(p.WmP) <= 160 * 1024 && (size_t)p.N * p.TC * 4 < ((size_t)1 << 32);
}

// coordinate kernel (WxP / 256 column shares per tile), then the message kernel
int launch_edge_bf16x3(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + kR - 1) / kR;
  hipLaunchKernelGGL(edge_x3_kernel<false>, dim3(tiles * (p.WxP / 256)), dim3(kT), x3_smem_bytes(p.WxP), st, p);
  hipLaunchKernelGGL(edge_x3_kernel<true>, dim3(tiles), dim3(kT), x3_smem_bytes(p.WmP), st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace egnn
