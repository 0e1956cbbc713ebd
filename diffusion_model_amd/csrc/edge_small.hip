// Edge kernels for SMALL graphs -- the reference's real per-call workload: generate() evaluates the network on ONE 20-64-atom
// graph per reverse step (parts/train_per_iretation.py:288-364), i.e. ~4 k edges.  On the 128-edge tiles of edge_x_m16.hip /
// edge_bf16_v4.hip such a layer is 64 + 32 workgroups on 256 CUs, each walking the whole K = 1024 for 128 rows: ~30 us of serial
// latency per layer with three quarters of the chip idle (bench.py `latency` leg).  Here a tile is 32 edges, so one 64-atom
// graph spreads over 126 x (2 + 1) workgroups and a workgroup's time is set by what it cannot avoid: streaming its columns'
// weight fragments from L2 (coordinate kernel: 512 columns x 1024 x 2 B = 1 MiB at the per-CU L2 path's 64 B/clk = 7 us)
// instead of by 128 rows of MFMA + SiLU work.  Same arithmetic as the large-tile kernels (EquivariantGraphNeuralNetwork.py
// :55-65: fp16 first-layer table, SiLU as t * rcp(1 + exp2(t)) on pre-scaled arguments, bf16 or fp16 MFMA operands with fp32
// accumulation, fp32 heads and segment sums, tile partials added by node_post in tile order: deterministic, no atomics);
// the K loop is the plain double-buffered form (build chunk c + 1 | multiply chunk c | one barrier) with the weight fragments
// requested two chunks ahead: at 32 rows there is no matrix / vector balance to tune, the loop waits for weights.
//
//   workgroup = 8 wave64, v_mfma_f32_16x16x32_{bf16,f16}: 2 row blocks x CBW column blocks of 16 per wave
//     IS_M = false  coordinate branch (:62-65): 512 columns of mlp_x.2 per workgroup (WxP / 512 column shares per tile), CBW = 4
//     IS_M = true   message branch (:57-61): all 256 columns of mlp_m.2 + the attention gate, CBW = 2
//   LDS activation image [8 k-groups][32 rows][16 B], rows XOR-swizzled by the k-group as in edge_x_m16.hip.
#include "diag.h"
#include "edge_tile.h"

namespace egnn {

namespace {

constexpr int kTS = 512;
constexpr int kKCS = 64;         // activation chunk depth
constexpr int kMLd = 257;        // row stride (floats) of the message tile [R][256] (odd: column walks hit 32 banks)
// LDS carve (bytes) for a tile of R edges: per-row arrays (19 x R words), then the activation images, wd[KP], the message tile
template <int R> struct SmallLds {
  static constexpr size_t kA1S = (size_t)8 * R * 16;   // one activation chunk image [8 k-groups][R rows][16 B]
  static constexpr size_t kSOffDst = 0, kSOffSrc = kSOffDst + R * 4, kSOffD2 = kSOffSrc + R * 4, kSOffDiff = kSOffD2 + R * 4,
                          kSOffVal = kSOffDiff + 3 * R * 4, kSOffPart = kSOffVal + R * 4, kSOffSegRow = kSOffPart + 8 * R * 4,
                          kSOffSegNode = kSOffSegRow + R * 4, kSOffSegRs = kSOffSegNode + R * 4, kSOffSegRe = kSOffSegRs + R * 4,
                          kSOffSegMode = kSOffSegRe + R * 4, kSOffMisc = kSOffSegMode + R * 4, kSOffA1 = kSOffMisc + 64;
  static __host__ __device__ size_t bytes(int KP, bool is_m) { return kSOffA1 + 2 * kA1S + (size_t)KP * 4 + (is_m ? ((size_t)R * kMLd + 256) * 4 : 0); }
};

// kRS = edges per tile.  32 is what is launched; the 64-row instantiation was measured for ONE 64-atom graph (4,032 edges) and is
// no faster than the 128-row kernels there (coordinate kernel 32.2 vs 28.8 us, message kernel 26.0 vs 23.8 us per layer,
// profiles/r04h_latency.log): at that size every kernel of a layer -- whatever its tiling -- takes ~25 us, the length of its
// chain of dependent memory round trips at the clocks the chip holds between tiny kernels, not its arithmetic.
template <bool IS_M, typename V8, int kRS>
__global__ __launch_bounds__(kTS, 1) void edge_small_kernel(const EdgeParams p) {
  if constexpr (OpTraits<V8>::f16) f16_saturate_mode();
  constexpr int CBW = IS_M ? 2 : 4;                      // 16-column blocks per wave
  constexpr int RB = kRS / 16;                           // 16-row blocks per tile
  constexpr float kAcc = kNegLog2e / OpTraits<V8>::wscale;
  typedef SmallLds<kRS> LY;
  constexpr size_t kA1S = LY::kA1S, kSOffDst = LY::kSOffDst, kSOffSrc = LY::kSOffSrc, kSOffD2 = LY::kSOffD2, kSOffDiff = LY::kSOffDiff,
                   kSOffVal = LY::kSOffVal, kSOffPart = LY::kSOffPart, kSOffSegRow = LY::kSOffSegRow, kSOffSegNode = LY::kSOffSegNode,
                   kSOffSegRs = LY::kSOffSegRs, kSOffSegRe = LY::kSOffSegRe, kSOffSegMode = LY::kSOffSegMode, kSOffMisc = LY::kSOffMisc,
                   kSOffA1 = LY::kSOffA1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* const s_dst = reinterpret_cast<int*>(smem + kSOffDst);
  int* const s_src = reinterpret_cast<int*>(smem + kSOffSrc);
  float* const s_d2 = reinterpret_cast<float*>(smem + kSOffD2);
  float* const s_diff = reinterpret_cast<float*>(smem + kSOffDiff);
  float* const s_val = reinterpret_cast<float*>(smem + kSOffVal);
  float* const s_part = reinterpret_cast<float*>(smem + kSOffPart);
  int* const s_seg_of_row = reinterpret_cast<int*>(smem + kSOffSegRow);
  int* const s_seg_node = reinterpret_cast<int*>(smem + kSOffSegNode);
  int* const s_seg_rs = reinterpret_cast<int*>(smem + kSOffSegRs);
  int* const s_seg_re = reinterpret_cast<int*>(smem + kSOffSegRe);
  int* const s_seg_mode = reinterpret_cast<int*>(smem + kSOffSegMode);
  int* const s_misc = reinterpret_cast<int*>(smem + kSOffMisc);
  char* const s_a1 = smem + kSOffA1;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r15 = lane & 15, q4 = lane >> 4;
  const int KP = IS_M ? p.WmP : p.WxP;
  float* const s_wd = reinterpret_cast<float*>(s_a1 + 2 * kA1S);
  float* const s_mt = s_wd + KP;                          // message tile (IS_M)
  const int nsplit = IS_M ? 1 : p.WxP / 512;
  const int j = blockIdx.x;
  const int tile = j / nsplit, half = j - tile * nsplit;
  const int e0 = tile * kRS;
  const int nvalid = min(kRS, p.E - e0);

  DIAG_STAMP_SETUP(p.stamps + ((size_t)(IS_M ? 1 : 0) * 8 + wave) * 32 * 4);
  DIAG_STAMP(30, 0);   // kernel entry
  // ---- weights of the first chunk: they depend on nothing, requested before anything else ----
  const int NC = KP / kKCS, KS = KP / 32;
  const rsrc_t rs_w = make_rsrc(IS_M ? p.w2m : p.w2x, (unsigned)((size_t)(IS_M ? p.MP : p.WxP) * KP * 2));
  const int cb0 = IS_M ? wave * CBW : half * 32 + wave * CBW;   // first 16-column block of this wave
  const unsigned w0 = (unsigned)cb0 * KS * 1024u, lane16 = lane * 16u;
  // two register sets of weight fragments: chunk c is multiplied from set c % 2, which is re-requested for chunk c + 2 as soon as
  // its MFMAs were issued (a third set made the 64-row coordinate kernel spill once both wave orders were instantiated)
  V8 ws0[2][CBW], ws1[2][CBW];
  auto wload = [&](V8 (&w)[2][CBW], const int c) {
    const unsigned ks = (unsigned)(c * 2) * 1024u;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb) w[s][cb] = ldbuf_v8<V8>(rs_w, lane16, w0 + (unsigned)cb * KS * 1024u + ks + (unsigned)s * 1024u);
  };
  wload(ws0, 0);
  if (NC > 1) wload(ws1, 1);

  // ---- prologue: edge rows, geometry (:56), segment structure of the CSR tile ----
  const float* wd = IS_M ? p.wdm : p.wdx;
  for (int i = tid; i < KP; i += kTS) s_wd[i] = wd[i];
  bool is_start = false, is_end = false;
  if (tid < kRS) {
    int d = 0, s = 0;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (tid < nvalid) {
      d = p.edge_dst[e0 + tid];
      s = p.edge_src[e0 + tid];
      dx = p.x[3 * d] - p.x[3 * s];
      dy = p.x[3 * d + 1] - p.x[3 * s + 1];
      dz = p.x[3 * d + 2] - p.x[3 * s + 2];
    }
    s_dst[tid] = d; s_src[tid] = s;
    s_diff[tid] = dx; s_diff[kRS + tid] = dy; s_diff[2 * kRS + tid] = dz;
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    s_d2[tid] = nrm * nrm;
    // consecutive rows with the same receiving node form a segment (the first kRS lanes of wave 0: neighbours by shuffle)
    const int dp = __shfl_up(d, 1), dn = __shfl_down(d, 1);
    const bool valid = tid < nvalid;
    is_start = valid && (tid == 0 || dp != d);
    is_end = valid && (tid == nvalid - 1 || dn != d);
    const unsigned long long starts = __ballot(is_start);
    const int seg = __popcll(starts & ((2ull << lane) - 1ull)) - 1;
    s_seg_of_row[tid] = valid ? seg : -1;
    if (is_start) { s_seg_node[seg] = d; s_seg_rs[seg] = tid; }
    if (is_end) s_seg_re[seg] = tid;
    if (tid == 0) s_misc[0] = __popcll(starts);
  }
  __syncthreads();
  DIAG_STAMP(30, 1);   // edge rows, geometry, segments ready
  const int S = s_misc[0];
  int my_mode = 0;
  if (tid < S) {   // 2 = all edges of the node are in this segment, 1 = the node's edges start here, 0 = continue (edge_tile.h)
    const int n = s_seg_node[tid];
    const bool first = (e0 + s_seg_rs[tid]) == p.row_ptr[n];
    const bool last = (e0 + s_seg_re[tid] + 1) == p.row_ptr[n + 1];
    my_mode = (first && last) ? 2 : (first ? 1 : 0);
  }

  // ---- K loop ----
  // build: threads 0..255 finish one unit (8 hidden units of one row) per chunk: SiLU(P[dst] + Q[src] + wd * d2) -> operand type
  const int brow = (tid >> 3) & (kRS - 1), kg = tid & 7;
  const rsrc_t rs_tab = make_rsrc(p.table, (unsigned)((size_t)p.N * p.TC * 2));
  const unsigned vdst = (unsigned)s_dst[brow] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const unsigned vsrc = (unsigned)s_src[brow] * (unsigned)p.TC * 2u + (unsigned)kg * 16u;
  const float d2r = s_d2[brow];
  const unsigned offP = (IS_M ? 2u * p.WxP : 0u) * 2u, offQ = (IS_M ? 2u * p.WxP + p.WmP : (unsigned)p.WxP) * 2u;   // fp16 table {Px|Qx|Pm|Qm}
  char* const slot = s_a1 + (size_t)kg * (kRS * 16) + (size_t)(brow ^ kg) * 16;
  const char* const afrag0 = s_a1 + (size_t)q4 * (kRS * 16) + (size_t)(r15 ^ q4) * 16;            // k-step 0: k-group q4
  const char* const afrag1 = s_a1 + (size_t)(4 + q4) * (kRS * 16) + (size_t)(r15 ^ (4 + q4)) * 16;   // k-step 1: k-group 4 + q4

  f32x4 acc[RB][CBW];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // first-layer table rows: two register sets, chunk k in set k % 2, re-requested for chunk k + 2 the moment chunk k was built
  // (a handful of workgroups cannot hide a memory latency behind each other, only behind their own earlier requests)
  UnitH u0, u1;
  auto tload = [&](UnitH& u, const int c) {
    const int cc = c < NC ? c : NC - 1;   // past the end: a harmless repeat
    unith_load(u, rs_tab, vdst, vsrc, offP + (unsigned)cc * kKCS * 2u, offQ + (unsigned)cc * kKCS * 2u);
  };
  // (every thread builds: at 32 rows waves 4-7 mirror waves 0-3 -- same units, same values written twice -- so that the loop
  // below has no branch: behind a control-flow join hipcc waits for EVERY outstanding load (vmcnt(0)), i.e. also for the weight
  // fragments requested a moment ago, which cost one L2 round trip per chunk: 2.8 k cycles per chunk against 0.9 k of MFMAs
  // in the first build, tools/stamps.py)
  tload(u0, 0); tload(u1, 1);
  unith_finish<V8>(u0, s_wd + kg * 8, d2r, slot);
  tload(u0, 2);
  if (tid < S) s_seg_mode[tid] = my_mode;
  __syncthreads();
  DIAG_STAMP(30, 2);   // chunk 0 built
  DIAG_RSTAMP(31, 1);
  // step c: multiply chunk c (weights wc), request the weights of chunk c + 2 (into wn2), finish the activations of chunk c + 1
  // from the table set ut and re-request that set for chunk c + 4.  No branch inside (clamped repeats past the end).
  // The two waves of a SIMD (w and w + 4) run a step in OPPOSITE order -- multiply chunk c then build chunk c + 1, or build then
  // multiply -- so that one's vector work (8 SiLUs per thread and chunk) runs under the other's matrix work, as in
  // edge_x_m16.hip: with all eight waves in the same phase a chunk cost build (both waves of a SIMD in turn, 1.8 k cycles) PLUS
  // MFMAs (1.0 k), tools/stamps.py.
  auto step = [&](const int c, V8 (&wc)[2][CBW], UnitH& ut, const bool mfma_first) {
    const size_t boff = (size_t)(c & 1) * kA1S, noff = (size_t)((c + 1) & 1) * kA1S;
    DIAG_STAMP(c, 0);
    const int cn = c + 1 < NC ? c + 1 : NC - 1;
    auto build = [&]() {   // activations of chunk c + 1 from the table set ut, which is then re-requested for chunk c + 3
      unith_finish<V8>(ut, s_wd + cn * kKCS + kg * 8, d2r, slot + noff);   // (last step: rebuilds the last chunk into the free buffer)
      tload(ut, c + 3);
    };
    auto multiply = [&]() {   // chunk c from the weight set wc, which is then re-requested for chunk c + 2
      // operand pieces of a row block (both k-steps) are read one row block ahead of their MFMAs: read right in front of their
      // use, every group of 4 MFMAs waited out an LDS round trip (1.3 k cycles per chunk for 0.5 k of matrix work)
      V8 a0 = *reinterpret_cast<const V8*>(afrag0 + boff), a1 = *reinterpret_cast<const V8*>(afrag1 + boff);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        V8 n0 = a0, n1 = a1;
        if (rb + 1 < RB) {
          n0 = *reinterpret_cast<const V8*>(afrag0 + boff + (rb + 1) * 256);   // rows 16 rb + r15 (the XOR only touches the low 3 bits)
          n1 = *reinterpret_cast<const V8*>(afrag1 + boff + (rb + 1) * 256);
        }
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) acc[rb][cb] = mfma16(a0, wc[0][cb], acc[rb][cb]);
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) acc[rb][cb] = mfma16(a1, wc[1][cb], acc[rb][cb]);
        a0 = n0; a1 = n1;
      }
      wload(wc, c + 2 < NC ? c + 2 : NC - 1);              // (past the end: harmless repeats instead of branches)
    };
    if (mfma_first) { multiply(); DIAG_STAMP(c, 1); __builtin_amdgcn_sched_barrier(0); build(); }
    else { build(); DIAG_STAMP(c, 1); __builtin_amdgcn_sched_barrier(0); multiply(); }
    DIAG_STAMP(c, 2);
    __syncthreads();
    DIAG_STAMP(c, 3);
  };
  // The register sets rotate statically, two steps per loop iteration, and the FIRST iteration is peeled: hipcc's wait insertion
  // joins the load counts of the two ways into the loop header, and with the prologue's request pattern on one side the first
  // step of EVERY iteration waited for almost every outstanding load (s_waitcnt vmcnt(1) in front of the activation build);
  // behind a peeled iteration both ways in carry the same pattern.  (NC even and >= 4: edge_small_supported.)
  auto kloop = [&](const bool mf) {
    step(0, ws0, u1, mf); step(1, ws1, u0, mf);
    for (int c = 2; c + 2 <= NC; c += 2) { step(c, ws0, u1, mf); step(c + 1, ws1, u0, mf); }
  };
  if (wave < 4) kloop(true);
  else kloop(false);

  DIAG_STAMP(30, 3);   // K loop done
  DIAG_RSTAMP(31, 2);
  // ---- epilogue ----
  // accumulator layout of a 16x16 tile: column = lane & 15, row = 4 (lane >> 4) + register
  if constexpr (!IS_M) {
    // s[row] = [b3] + sum over this workgroup's 512 columns of w3[n] SiLU(a2[row][n] + b2[n])   (:62-63)
    float part[RB * 4];
#pragma unroll
    for (int v = 0; v < RB * 4; ++v) part[v] = 0.f;
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) {
      const int n = 16 * (cb0 + cb) + r15;
      const float bb = p.b2x[n], w = p.w3x[n];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[rb * 4 + i] = fmaf(w, silu_s(fmaf(acc[rb][cb][i], kAcc, bb)), part[rb * 4 + i]);
    }
#pragma unroll
    for (int v = 0; v < RB * 4; ++v) {
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) part[v] += __shfl_xor(part[v], m);   // over the 16 columns of the lane group
    }
    if (r15 == 0) {
#pragma unroll
      for (int v = 0; v < RB * 4; ++v) s_part[wave * kRS + 16 * (v >> 2) + 4 * q4 + (v & 3)] = part[v];
    }
    __syncthreads();
    if (tid < kRS) {
      float v = half == 0 ? p.scal[0] : 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += s_part[w * kRS + tid];
      s_val[tid] = v;
    }
    __syncthreads();
    // coordinate messages (:64-65): per segment sum of (x_i - x_j) s_ij; component 3 = the segment's sum of |x_i - x_j|^2
    float* aggx = p.agg_x + (size_t)half * p.agg_x_stride;
    float* partx = p.part_x + (size_t)half * p.part_x_stride;
    if (S <= 8) {
      // few segments (a dense graph: 1-3 receiving nodes per tile): wave d < 4 owns component d, its lanes the rows, one
      // cross-lane sum per segment (a serial walk over a segment's rows was 9 k of the first build's 16 k epilogue cycles)
      if (wave < 4) {
        const int d = wave, row = lane;
        float v = 0.f;
        int sg = -1;
        if (row < kRS && row < nvalid) {
          sg = s_seg_of_row[row];
          if (d < 3) v = s_diff[d * kRS + row] * s_val[row];
          else { const float dx = s_diff[row], dy = s_diff[kRS + row], dz = s_diff[2 * kRS + row]; v = dx * dx + dy * dy + dz * dz; }
        }
        for (int seg = 0; seg < S; ++seg) {
          float t = sg == seg ? v : 0.f;
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m);
          if (lane == 0) {
            const int mode = s_seg_mode[seg];
            float* dstp = mode == 2 ? aggx + (size_t)s_seg_node[seg] * 4 : partx + ((size_t)tile * 2 + mode) * 4;
            dstp[d] = t;
          }
        }
      }
    } else if (tid < 4 * S) {
      const int seg = tid >> 2, d = tid & 3, mode = s_seg_mode[seg];
      float sum = 0.f;
      for (int rr = s_seg_rs[seg]; rr <= s_seg_re[seg]; ++rr) {
        if (d < 3) sum += s_diff[d * kRS + rr] * s_val[rr];
        else {
          const float dx = s_diff[rr], dy = s_diff[kRS + rr], dz = s_diff[2 * kRS + rr];
          sum += dx * dx + dy * dy + dz * dz;
        }
      }
      float* dstp = mode == 2 ? aggx + (size_t)s_seg_node[seg] * 4 : partx + ((size_t)tile * 2 + mode) * 4;
      dstp[d] = sum;
    }
  } else {
    // m = SiLU(a2 + b2), gate = sigmoid(wa . m + ba), messages m * gate summed per receiving node   (:57-61)
    float mval[RB][CBW][4];
    float zp[RB * 4];
#pragma unroll
    for (int v = 0; v < RB * 4; ++v) zp[v] = 0.f;
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) {
      const int n = 16 * (cb0 + cb) + r15;
      const float bb = p.b2m[n], wa = p.wa[n];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float m = silu_s(fmaf(acc[rb][cb][i], kAcc, bb));   // = -log2(e) * m
          mval[rb][cb][i] = m;
          zp[rb * 4 + i] = fmaf(wa, m, zp[rb * 4 + i]);
        }
    }
#pragma unroll
    for (int v = 0; v < RB * 4; ++v) {
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) zp[v] += __shfl_xor(zp[v], m);
    }
    if (r15 == 0) {
#pragma unroll
      for (int v = 0; v < RB * 4; ++v) s_part[wave * kRS + 16 * (v >> 2) + 4 * q4 + (v & 3)] = zp[v];
    }
    __syncthreads();
    if (tid < kRS) {
      float g = p.scal[1];
#pragma unroll
      for (int w = 0; w < 8; ++w) g += s_part[w * kRS + tid];
      s_val[tid] = sigmoid_f(g) * kNegInvLog2e;   // also undoes the scale of mval
    }
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 16 * rb + 4 * q4 + i;
        const float g = s_val[row];
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) s_mt[row * kMLd + 16 * (cb0 + cb) + r15] = mval[rb][cb][i] * g;
      }
    __syncthreads();
    // two half-columns of threads per message column (tid and tid + 256 take alternate rows of a segment), segment by segment:
    // the row reads of a segment are independent LDS reads (a walk that looked the segment up per row was a chain of dependent
    // LDS round trips: 20 k cycles of the first build's message epilogue)
    {
      const int col = tid & 255, par = tid >> 8;
      float* const s_half = s_mt + (size_t)kRS * kMLd;   // [256] sums of the odd rows (the tile itself is no longer needed there)
      for (int seg = 0; seg < S; ++seg) {
        const int rs = s_seg_rs[seg], re = s_seg_re[seg];
        float sum = 0.f;
#pragma unroll 8
        for (int rr = rs + par; rr <= re; rr += 2) sum += s_mt[rr * kMLd + col];
        if (seg > 0) __syncthreads();                  // the previous segment's exchange has been read
        if (par == 1) s_half[col] = sum;
        __syncthreads();
        if (par == 0 && col < p.MP) {
          const int mode = s_seg_mode[seg];
          float* dstp = mode == 2 ? p.agg_m + (size_t)s_seg_node[seg] * p.MP : p.part_m + ((size_t)tile * 2 + mode) * p.MP;
          dstp[col] = sum + s_half[col];
        }
      }
    }
  }
  DIAG_STAMP(31, 0);   // epilogue done
}

template <bool IS_M, typename V8, int R>
int launch_small(const EdgeParams& p, hipStream_t st) {
  const int tiles = (p.E + R - 1) / R;
  const int nsplit = IS_M ? 1 : p.WxP / 512;
  static bool attr_done = false;   // (per instantiation) the 64-row message kernel needs more than the default 64 KiB of LDS
  if (!attr_done) {
    EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_small_kernel<IS_M, V8, R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL((edge_small_kernel<IS_M, V8, R>), dim3(tiles * nsplit), dim3(kTS), SmallLds<R>::bytes(IS_M ? p.WmP : p.WxP, IS_M), st, p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace

// the shapes of edge_x_m16.hip / edge_bf16_v4.hip (hidden width 512 / 1024, 256 message columns), 16-column fragment streams packed
bool edge_small_supported(const EdgeParams& p) {
  return (p.WxP == 512 || p.WxP == 1024) && p.MP == 256 && p.WmP % 128 == 0 && p.WmP >= 256 && p.w2x16 != nullptr && p.w2m16 != nullptr &&
         (size_t)p.N * p.TC * 2 < ((size_t)1 << 32);
}

// p.w2x16 / p.w2m16 = the 16-column fragment streams of the chosen operand type (scaled); rows = 32 or 64 edges per tile
int launch_edge_small_x(const EdgeParams& p, hipStream_t st, bool f16, int rows) {
  EdgeParams q = p;
  q.w2x = p.w2x16;
  if (rows == 64) return f16 ? launch_small<false, f16x8, 64>(q, st) : launch_small<false, bf16x8, 64>(q, st);
  return f16 ? launch_small<false, f16x8, 32>(q, st) : launch_small<false, bf16x8, 32>(q, st);
}
int launch_edge_small_m(const EdgeParams& p, hipStream_t st, bool f16, int rows) {
  EdgeParams q = p;
  q.w2m = p.w2m16;
  if (rows == 64) return f16 ? launch_small<true, f16x8, 64>(q, st) : launch_small<true, bf16x8, 64>(q, st);
  return f16 ? launch_small<true, f16x8, 32>(q, st) : launch_small<true, bf16x8, 32>(q, st);
}

}  // namespace egnn
