// node_post on bf16 MFMA (throughput mode): h' = mlp_h([h | sum_m]) (EquivariantGraphNeuralNetwork.py:26-30, :69)
// and x' = x + sum_x / (G + 1) (:64, :70) for 32 nodes per workgroup.
//
// Both GEMMs run TRANSPOSED (nodes on the MFMA lanes): hidden^T = W1h . X^T leaves the hidden units in the
// accumulator registers and the node on the lane, which is exactly the B-operand layout of the second product
// out^T = W2h . hidden^T, so the 1024-wide hidden activation never touches LDS or HBM: bias + SiLU + bf16 pack
// happen on the accumulator registers (the k order inside a 16-deep step is the accumulator's row order; W2h is
// packed with the same permutation).  Each wave owns a quarter of the hidden units; the four partial out^T
// tiles are added through LDS in a fixed order.
#include <stdlib.h>

#include "diag.h"
#include "kernels.h"

namespace egnn {

namespace {

constexpr int kNodes = 32, kThreadsN = 256;
constexpr int kMaxKS1 = 20;   // k-steps of the first product held in registers (H + MP <= 320)
constexpr int kSplitK = 320;  // K of the split-operand form: 20 k-steps = 2 turns of the 10-deep weight ring (H + MP in (160, 320])
constexpr int kRingD = 10;
__host__ __device__ inline size_t nb_smem_bytes(int K1Q, int OB, bool split = false) {
  const size_t xs = (size_t)(K1Q / 8) * 33 * 16 * (split ? 2 : 1);   // X^T image [k-group][33][8 bf16] (split: head + remainder)
  const size_t red = (size_t)4 * OB * 16 * 64 * 4;         // cross-wave partial out^T
  return xs + red;
}

// OBT = compile-time bound of the output column blocks (H <= 32 OBT).  The accumulators of absent blocks cost registers:
// with a single H <= 256 instantiation the kernel took 464 registers per lane = ONE workgroup per CU, and a layer's 512
// workgroups ran as two serial rounds of a latency-bound chain (88-98 us); at OBT = 2 (H <= 64, 224 VGPRs) two workgroups
// share a CU: 49 us.
// V8 = the MFMA operand type: bf16x8, or f16x8 for precision fp16 (p.w1h_bf16 / p.w2h_bf16p then point at the fp16 fragment
// streams, packed x 2^8: both accumulators are divided by it where the biases are added).
// SPLIT = true (fp32-grade node MLP on the half-precision matrix cores; used by precision fp16 and bf16x3): every operand of
// both products is a head + remainder pair (X = X_hi + X_lo, W = W_hi + W_lo, hidden = h_hi + h_lo; with fp16 pairs 22
// significant bits) and each product is three MFMAs, small terms first: lo x hi + hi x lo + hi x hi.  The node MLP is the
// largest single rounding-error source of the fp16 path (tools/rounding_budget.py: 4.5e-4 of 5.3e-4 on h', 6.2e-4 of 8.3e-4
// on eps_x) and costs 0.4 % of a layer's FLOP.  W1h head / remainder fragments stream through a ring of kRingD k-steps
// (requested kRingD k-steps ahead, across hidden-block boundaries) instead of living in registers for a whole block;
// K is padded to kSplitK = 2 ring turns so that the ring slot of a k-step is a compile-time constant.
// HS = true: the hidden-split launch of small graphs (8 workgroups per node tile, ONE hidden block per wave; a layer is then a
// handful of workgroups and the kernel's duration is the length of its chains of dependent memory round trips): no
// next-block prefetch (there is no next block; its 80 registers go to the gather), and the gather of [h | sum_m] is BRANCH-FREE:
// one or two unconditional loads per element from a selected address (the node's own slot or the first two tile partials),
// 20 loads of a thread in flight at once (four batches; all 80 at once spill).  With `if (k < H) ... else if (t0 == t1) ...` around the loads every iteration
// was a control-flow join, behind which hipcc waits for all outstanding loads: 19 serial round trips, 12 of the kernel's 27 us
// at one 64-atom graph (timing builds EGNN_EXP_NP_*, profiles/r04h_latency.log).
template <int OBT, typename V8 = bf16x8, bool SPLIT = false, bool HS = false>
__global__ __launch_bounds__(kThreadsN, (OBT > 2 ? 1 : 2)) void node_post_bf16_kernel(const PostParams p) {
  typedef typename OpTraits<V8>::elem elem;
  constexpr float kInvW = 1.0f / OpTraits<V8>::wscale;
  if constexpr (OpTraits<V8>::f16) f16_saturate_mode();
  constexpr bool PF = !HS;   // the W1h fragments of the next hidden block are requested while this one is multiplied
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int n0 = blockIdx.x * kNodes;
  const int K1Q = SPLIT ? kSplitK : p.K1Q;
  const int KS1 = K1Q / 16, OB = p.HP / 32;
  char* Xl = Xb + (size_t)(K1Q / 8) * 33 * 16;   // SPLIT: remainder image
  float* red = reinterpret_cast<float*>(smem + (size_t)(K1Q / 8) * 33 * 16 * (SPLIT ? 2 : 1));

  const V8* w1 = reinterpret_cast<const V8*>(p.w1h_bf16);    // [WhP/32][KS1][64]
  const V8* w2 = reinterpret_cast<const V8*>(p.w2h_bf16p);   // [OB][WhP/16][64], k permuted
  const int KS2 = p.WhP / 16;
  // hidden units: split over the gridDim.y workgroups of a node tile (1 = all here), then over the 4 waves
  const int hsplit = gridDim.y, hsi = blockIdx.y;
  const int nhb = p.WhP / 32, hb_per_wave = nhb / (4 * hsplit), hb0 = hsi * (nhb / hsplit);
  // the first hidden block's W1h fragments depend on nothing: requested before the gather so that their L2 / HBM round
  // trip runs under it (at B = 1 a layer's node_post is a chain of such round trips and little else)
  const V8* w1l = reinterpret_cast<const V8*>(p.w1h_lo);    // SPLIT: remainders, same layouts
  const V8* w2l = reinterpret_cast<const V8*>(p.w2h_lo);
  V8 wf[SPLIT ? 1 : kMaxKS1];
  V8 rh[SPLIT ? kRingD : 1], rl[SPLIT ? kRingD : 1];   // SPLIT: ring of W1h head / remainder fragments
  if constexpr (SPLIT) {
    const size_t o = ((size_t)(hb0 + wave * hb_per_wave) * (kSplitK / 16)) * 64 + lane;
#pragma unroll
    for (int s = 0; s < kRingD; ++s) { rh[s] = w1[o + (size_t)s * 64]; rl[s] = w1l[o + (size_t)s * 64]; }
  } else {
    const V8* w1b = w1 + ((size_t)(hb0 + wave * hb_per_wave) * KS1) * 64 + lane;
#pragma unroll
    for (int s = 0; s < kMaxKS1; ++s) wf[s] = w1b[(size_t)(s < KS1 ? s : 0) * 64];
  }
  // the same for the first hidden block's biases and its W2h fragments (two more round trips of the B = 1 chain)
  float b1v[16];
  V8 w2f[OBT][2];
  V8 w2g[SPLIT ? OBT : 1][2];   // SPLIT: remainders of the W2h fragments
  {
    const int hbf = hb0 + wave * hb_per_wave;
#pragma unroll
    for (int i = 0; i < 16; ++i) b1v[i] = p.b1h[32 * hbf + acc_row(i, lane)];
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        w2f[ob][s] = w2[((size_t)(ob < OB ? ob : 0) * KS2 + 2 * hbf + s) * 64 + lane];
        if constexpr (SPLIT) w2g[ob][s] = w2l[((size_t)(ob < OB ? ob : 0) * KS2 + 2 * hbf + s) * 64 + lane];
      }
  }

  // gather [h | sum_m] (tile partials added in tile order), pack to bf16 in the fragment image.  The CSR lookups of the
  // 32 nodes go through LDS first and the item loop is unrolled: independent loads in flight instead of a chain of
  // dependent ones per item (which nothing hides when the layer has one or two of these workgroups).
  int* s_t0 = reinterpret_cast<int*>(red);          // [kNodes] first tile of the node's edges, -1 = no edges
  int* s_t1 = s_t0 + kNodes;                        // [kNodes] last tile
  if (tid < kNodes) {
    const int n = n0 + tid;
    int t0 = -1, t1 = -1;
    if (n < p.N) {
      const int rp0 = p.row_ptr[n], rp1 = p.row_ptr[n + 1];
      if (rp1 > rp0) { t0 = rp0 / p.R; t1 = (rp1 - 1) / p.R; }
    }
    s_t0[tid] = t0; s_t1[tid] = t1;
  }
  __syncthreads();
  if constexpr (HS) {
    constexpr int kIt = (kNodes * (kMaxKS1 * 16 / 2) + kThreadsN - 1) / kThreadsN;   // 20 pairs per thread at K1Q = 320
    constexpr int kBatch = 5;                                                          // pairs whose loads fly together
    static_assert(kIt % kBatch == 0, "batches of equal size");
#pragma unroll
    for (int b0 = 0; b0 < kIt; b0 += kBatch) {
      float va[kBatch][2], vb[kBatch][2];
      bool more = false;
#pragma unroll
      for (int it = 0; it < kBatch; ++it) {
        const int i = min(tid + kThreadsN * (b0 + it), kNodes * (K1Q / 2) - 1);
        const int node = i / (K1Q / 2), kp = i % (K1Q / 2), n = min(n0 + node, p.N - 1);
        const int t0 = s_t0[node], t1 = s_t1[node];
        more |= t1 > t0 + 1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int k = 2 * kp + u, c = k - p.H;
          const bool is_h = k < p.H, is_m = !is_h && c < p.MP && t0 >= 0;
          const float* a = is_h ? p.h + (size_t)n * p.H + k
                                : (t0 == t1 ? p.agg_m + (size_t)n * p.MP + (is_m ? c : 0) : p.part_m + ((size_t)(is_m ? t0 : 0) * 2 + 1) * p.MP + (is_m ? c : 0));
          const bool two = is_m && t1 > t0;
          const float* b = two ? p.part_m + ((size_t)(t0 + 1) * 2) * p.MP + c : p.h;
          const float x0 = *a, x1 = *b;      // unconditional: no branch between the loads
          va[it][u] = (is_h || is_m) ? x0 : 0.f;
          vb[it][u] = two ? x1 : 0.f;
        }
      }
#pragma unroll
      for (int it = 0; it < kBatch; ++it) {
        const int i = tid + kThreadsN * (b0 + it);
        if (i < kNodes * (K1Q / 2)) {
          const int node = i / (K1Q / 2), kp = i % (K1Q / 2), n = n0 + node;
          float v[2] = {va[it][0] + vb[it][0], va[it][1] + vb[it][1]};
          if (n >= p.N) v[0] = v[1] = 0.f;
          else if (more) {   // a node whose edges span more than two tiles (degree > 2 R): the remaining partials
            const int t0 = s_t0[node], t1 = s_t1[node];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int c = 2 * kp + u - p.H;
              if (c >= 0 && c < p.MP)
                for (int t = t0 + 2; t <= t1; ++t) v[u] += p.part_m[((size_t)t * 2) * p.MP + c];
            }
          }
          const int k = 2 * kp;
          elem* dst = reinterpret_cast<elem*>(Xb + ((size_t)(k >> 3) * 33 + node) * 16) + (k & 7);
          const elem e0 = (elem)v[0], e1 = (elem)v[1];
          dst[0] = e0;
          dst[1] = e1;
          if constexpr (SPLIT) {
            elem* dl = reinterpret_cast<elem*>(Xl + ((size_t)(k >> 3) * 33 + node) * 16) + (k & 7);
            dl[0] = (elem)(v[0] - (float)e0);
            dl[1] = (elem)(v[1] - (float)e1);
          }
        }
      }
    }
  } else
#pragma unroll 4
  for (int i = tid; i < kNodes * (K1Q / 2); i += kThreadsN) {
    const int node = i / (K1Q / 2), kp = i % (K1Q / 2), n = n0 + node;
    float v[2] = {0.f, 0.f};
    if (n < p.N) {
      const int t0 = s_t0[node], t1 = s_t1[node];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = 2 * kp + u;
        if (k < p.H) v[u] = p.h[(size_t)n * p.H + k];
        else if (k - p.H < p.MP && t0 >= 0) {
          const int c = k - p.H;
          if (t0 == t1) v[u] = p.agg_m[(size_t)n * p.MP + c];
          else {
            float a = p.part_m[((size_t)t0 * 2 + 1) * p.MP + c];
            for (int t = t0 + 1; t <= t1; ++t) a += p.part_m[((size_t)t * 2) * p.MP + c];
            v[u] = a;
          }
        }
      }
    }
    const int k = 2 * kp;
    elem* dst = reinterpret_cast<elem*>(Xb + ((size_t)(k >> 3) * 33 + node) * 16) + (k & 7);
    const elem e0 = (elem)v[0], e1 = (elem)v[1];
    dst[0] = e0;
    dst[1] = e1;
    if constexpr (SPLIT) {
      elem* dl = reinterpret_cast<elem*>(Xl + ((size_t)(k >> 3) * 33 + node) * 16) + (k & 7);
      dl[0] = (elem)(v[0] - (float)e0);
      dl[1] = (elem)(v[1] - (float)e1);
    }
  }
  // normaliser G^2 = sum of d^2 over the edges of the node's graph (:64), from component 3 of the coordinate sums.
  // Small graphs (<= 64 nodes): 8 lanes per node.  Larger graphs: the whole workgroup sums each distinct graph of its
  // 32 nodes (a latency-bound loop of dependent loads otherwise).  Either way every workgroup that needs a graph's
  // sum adds the same values in the same order: the result is bitwise the same everywhere.
  float* gsq = reinterpret_cast<float*>(smem + nb_smem_bytes(K1Q, p.HP / 32, SPLIT));   // [kNodes] then [kThreadsN] scratch
  if (p.sq_from_agg && blockIdx.y == 0 && !diag::kNpNoNorm) {
    float* red = gsq + kNodes;
    const int node = tid >> 3, n = n0 + node;
    int g = -1, lo = 0, hi = 0;
    if (n < p.N) { g = p.node_graph[n]; lo = p.graph_ptr[g]; hi = p.graph_ptr[g + 1]; }
    if (g >= 0 && hi - lo <= 64) {
      const float sq = graph_sq_sum8(lo, hi, tid & 7, p.row_ptr, p.R, p.agg_x, p.part_x);
      if ((tid & 7) == 0) gsq[node] = sq;
    }
    const int nlast = min(n0 + kNodes, p.N) - 1;
    const int g_first = p.node_graph[n0], g_last = p.node_graph[nlast];
    for (int gg = g_first; gg <= g_last; ++gg) {          // workgroup-uniform loop over the distinct graphs
      const int glo = p.graph_ptr[gg], ghi = p.graph_ptr[gg + 1];
      if (ghi - glo <= 64) continue;
      float sacc = 0.f;
      for (int m = glo + tid; m < ghi; m += kThreadsN) sacc += node_sq_sum(m, p.row_ptr, p.R, p.agg_x, p.part_x);
      red[tid] = sacc;
      __syncthreads();
      for (int w = kThreadsN / 2; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
      }
      const float tot = red[0];
      __syncthreads();
      if (g == gg && (tid & 7) == 0) gsq[node] = tot;
    }
    __syncthreads();
  }
  // coordinate update (by the first of the workgroups that share this node tile)
  if (blockIdx.y == 0 && tid < kNodes * 3 && !diag::kNpNoNorm) {
    const int node = tid / 3, d = tid % 3, n = n0 + node;
    if (n < p.N) {
      const int t0 = s_t0[node], t1 = s_t1[node];
      // the column-split copies of the coordinate sums (at most 4): every load unconditional from a selected address -- the
      // node's own slot or the first two tile partials -- so that they fly together (a branch per copy was a serial round
      // trip per copy in the small-graph regime); more than two tiles per node (degree > 2 R) adds the rest in a loop
      const bool has = t0 >= 0, one = t0 == t1;
      const float xin = p.x[3 * n + d];
      float va[4], vb[4];
#pragma unroll
      for (int hs = 0; hs < 4; ++hs) {
        const bool on = has && hs < p.nsplit_x;
        const float* ax = p.agg_x + (size_t)(on ? hs : 0) * p.agg_x_stride;
        const float* px = p.part_x + (size_t)(on ? hs : 0) * p.part_x_stride;
        const float* a = one || !on ? ax + (size_t)n * 4 + d : px + ((size_t)t0 * 2 + 1) * 4 + d;
        const float* b = on && !one ? px + ((size_t)(t0 + 1) * 2) * 4 + d : ax + (size_t)n * 4 + d;
        const float x0 = *a, x1 = *b;
        va[hs] = on ? x0 : 0.f;
        vb[hs] = on && !one ? x1 : 0.f;
      }
      float v = 0.f;
#pragma unroll
      for (int hs = 0; hs < 4; ++hs) {
        float w = va[hs] + vb[hs];
        if (has && hs < p.nsplit_x && t1 > t0 + 1) {
          const float* px = p.part_x + (size_t)hs * p.part_x_stride;
          for (int t = t0 + 2; t <= t1; ++t) w += px[((size_t)t * 2) * 4 + d];
        }
        v += w;
      }
      const float sq = p.sq_from_agg ? gsq[node] : p.gscale[p.per_graph ? p.node_graph[n] : 0];
      const float g = 1.0f / (sqrtf(sq) + 1.0f);
      p.x_out[3 * n + d] = xin + v * g;
    }
  }
  __syncthreads();

  f32x16 oacc[OBT];
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[ob][i] = 0.f;

  const char* xfrag = Xb + ((size_t)hh * 33 + r) * 16;   // B fragment of k-step s: + s*2*33*16
  // The W1h fragments of a hidden block (KS1 <= kMaxKS1 k-steps, 1 KiB each) are all requested before the
  // block's MFMA chain, and the next block's while this one is multiplied: one exposed L2 round trip per
  // wave instead of one per k-step.
  V8 wn[SPLIT ? 1 : kMaxKS1];
  const char* xfrag_lo = Xl + ((size_t)hh * 33 + r) * 16;
  for (int q = 0; q < (diag::kNpNoMlp ? 0 : hb_per_wave); ++q) {
    const int hb = hb0 + wave * hb_per_wave + q;
    if constexpr (SPLIT) {
    } else if constexpr (PF) {
      const int hbn = q + 1 < hb_per_wave ? hb + 1 : hb;
      const V8* w1b = w1 + ((size_t)hbn * KS1) * 64 + lane;
#pragma unroll
      for (int s = 0; s < kMaxKS1; ++s) wn[s] = w1b[(size_t)(s < KS1 ? s : 0) * 64];
    } else if (q > 0) {
      const V8* w1b = w1 + ((size_t)hb * KS1) * 64 + lane;
#pragma unroll
      for (int s = 0; s < kMaxKS1; ++s) wf[s] = w1b[(size_t)(s < KS1 ? s : 0) * 64];
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if constexpr (SPLIT) {
      const int hbn = q + 1 < hb_per_wave ? hb + 1 : hb;   // (past the last block: a harmless repeat)
      const size_t oc = ((size_t)hb * (kSplitK / 16)) * 64 + lane, on = ((size_t)hbn * (kSplitK / 16)) * 64 + lane;
      static_assert((kSplitK / 16) % kRingD == 0, "the ring slot of a k-step must be a compile-time constant");
#pragma unroll 1
      for (int turn = 0; turn < kSplitK / 16 / kRingD; ++turn) {   // (not unrolled: bounds the LDS operands hipcc hoists)
        const bool last_turn = turn == kSplitK / 16 / kRingD - 1;
        // the k-step a freed slot is refilled with: kRingD ahead in this block, or the next block's first turn
        const size_t ofill = last_turn ? on : oc + (size_t)(turn + 1) * kRingD * 64;
#pragma unroll
        for (int u = 0; u < kRingD; ++u) {
          const int s = turn * kRingD + u;
          const V8 bh = *reinterpret_cast<const V8*>(xfrag + (size_t)s * 2 * 33 * 16);
          const V8 bl = *reinterpret_cast<const V8*>(xfrag_lo + (size_t)s * 2 * 33 * 16);
          acc = mfma32(rl[u], bh, acc);
          acc = mfma32(rh[u], bl, acc);
          acc = mfma32(rh[u], bh, acc);
          rh[u] = w1[ofill + (size_t)u * 64];
          rl[u] = w1l[ofill + (size_t)u * 64];
        }
      }
    } else {
#pragma unroll
    for (int s = 0; s < kMaxKS1; ++s)
      if (s < KS1) {
        const V8 b = *reinterpret_cast<const V8*>(xfrag + (size_t)s * 2 * 33 * 16);
        acc = mfma32(wf[s], b, acc);
      }
    }
    // bias + SiLU on the accumulator; registers 8s..8s+7 are the B fragment of k-step s of the second product
    V8 hf[2];
    V8 hg[SPLIT ? 2 : 1];   // SPLIT: remainders of the hidden activation
    if (q > 0) {   // (block 0: requested at the top)
#pragma unroll
      for (int i = 0; i < 16; ++i) b1v[i] = p.b1h[32 * hb + acc_row(i, lane)];
#pragma unroll
      for (int ob = 0; ob < OBT; ++ob)
        if (ob < OB) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            w2f[ob][s] = w2[((size_t)ob * KS2 + 2 * hb + s) * 64 + lane];
            if constexpr (SPLIT) w2g[ob][s] = w2l[((size_t)ob * KS2 + 2 * hb + s) * 64 + lane];
          }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float hv = silu_f(OpTraits<V8>::f16 ? fmaf(acc[i], kInvW, b1v[i]) : acc[i] + b1v[i]);
      const elem he = (elem)hv;
      hf[i >> 3][i & 7] = he;
      if constexpr (SPLIT) hg[i >> 3][i & 7] = (elem)(hv - (float)he);
    }
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob)
      if (ob < OB) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if constexpr (SPLIT) {
            oacc[ob] = mfma32(w2g[ob][s], hf[s], oacc[ob]);
            oacc[ob] = mfma32(w2f[ob][s], hg[s], oacc[ob]);
          }
          oacc[ob] = mfma32(w2f[ob][s], hf[s], oacc[ob]);
        }
      }
    if constexpr (PF && !SPLIT) {
#pragma unroll
      for (int s = 0; s < kMaxKS1; ++s) wf[s] = wn[s];
    }
  }
  // cross-wave sum (fixed order) and store: oacc[ob][i] = out^T[o = 32ob + acc_row(i), node = r]
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob)
    if (ob < OB) {
#pragma unroll
      for (int i = 0; i < 16; ++i) red[((size_t)(wave * OB + ob) * 16 + i) * 64 + lane] = oacc[ob][i];
    }
  __syncthreads();
  for (int e = tid; e < OB * 16 * 64; e += kThreadsN) {
    const int l = e & 63, i = (e >> 6) & 15, ob = e >> 10;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[((size_t)(w * OB + ob) * 16 + i) * 64 + l];
    if constexpr (OpTraits<V8>::f16) v *= kInvW;
    const int o = 32 * ob + acc_row(i, l), n = n0 + (l & 31);
    if (n < p.N && o < p.H) {
      if (hsplit == 1) p.h_out[(size_t)n * p.H + o] = v + p.b2h[o];
      else p.h_partial[((size_t)hsi * p.N + n) * p.H + o] = v;   // summed (+ bias) by node_post_finish_kernel
    }
  }
}

// second launch of the hidden-split form: h'[n][o] = b2[o] + sum over the splits, in split order (deterministic)
__global__ void node_post_finish_kernel(int N, int H, int hsplit, const float* __restrict__ partial,
                                        const float* __restrict__ b2h, float* __restrict__ h_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * H) return;
  float pv[8];   // (the split is at most 8-fold: all loads first, then the sum in split order)
#pragma unroll
  for (int s = 0; s < 8; ++s) pv[s] = partial[(size_t)(s < hsplit ? s : 0) * N * H + i];
  float v = b2h[i % H];
#pragma unroll
  for (int s = 0; s < 8; ++s) v += s < hsplit ? pv[s] : 0.f;
  h_out[i] = v;
}

}  // namespace

int init_node_bf16_attributes() {
  const void* fns[] = {reinterpret_cast<const void*>(&node_post_bf16_kernel<2>),
                       reinterpret_cast<const void*>(&node_post_bf16_kernel<kPostMaxOB>),
                       reinterpret_cast<const void*>(&node_post_bf16_kernel<2, f16x8>),
                       reinterpret_cast<const void*>(&node_post_bf16_kernel<kPostMaxOB, f16x8>),
                       reinterpret_cast<const void*>(&node_post_bf16_kernel<2, f16x8, true>),
                       reinterpret_cast<const void*>(&node_post_bf16_kernel<2, bf16x8, false, true>),
                       reinterpret_cast<const void*>(&node_post_bf16_kernel<2, f16x8, true, true>)};
  for (const void* f : fns) EGNN_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return EGNN_OK;
}

bool node_post_bf16_supported(const PostParams& q) {
  return q.w1h_bf16 && q.w2h_bf16p && q.WhP % 128 == 0 && q.HP / 32 <= kPostMaxOB && q.K1Q / 16 <= kMaxKS1 &&
         nb_smem_bytes(q.K1Q, q.HP / 32) + (kNodes + kThreadsN) * 4 <= 160 * 1024;
}

// split-operand form: H <= 64 (two output blocks), 160 < H + MP <= 320, head + remainder fp16 streams packed for K = kSplitK
bool node_post_split_supported(const PostParams& q) {
  return q.w1h_lo && q.w2h_lo && q.w1h_bf16 && q.w2h_bf16p && q.HP <= 64 && q.H + q.MP > kSplitK / 2 && q.H + q.MP <= kSplitK &&
         q.WhP % 128 == 0 && nb_smem_bytes(kSplitK, q.HP / 32, true) + (kNodes + kThreadsN) * 4 <= 80 * 1024;
}
int node_post_split_k() { return kSplitK; }

// f16: q.w1h_bf16 / q.w2h_bf16p are the fp16 fragment streams (precision fp16); split: + q.w1h_lo / q.w2h_lo (K = kSplitK)
int launch_node_post_finish(int N, int H, int hs, const float* partial, const float* b2h, float* h_out, hipStream_t st) {
  hipLaunchKernelGGL(node_post_finish_kernel, dim3((N * H + 255) / 256), dim3(256), 0, st, N, H, hs, partial, b2h, h_out);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int launch_node_post_bf16(const PostParams& q, hipStream_t st, bool f16, bool split, bool defer_finish, int* hs_out) {
  const int tiles = (q.N + kNodes - 1) / kNodes;
  // Few node tiles (small graphs): a layer would wait for ONE workgroup's serial chain over all hidden blocks (40 us).
  // Split the hidden units over `hs` workgroups per tile (partial h' to scratch) and add them up in a second tiny launch:
  // the kernel boundary is the synchronisation, no cross-workgroup fences.
  int hs = 1;
  if (q.h_partial && tiles * 8 <= 256 && (q.WhP / 32) % 32 == 0) hs = 8;
  const dim3 grid(tiles, hs);
  const size_t sm = nb_smem_bytes(split ? kSplitK : q.K1Q, q.HP / 32, split) + (kNodes + kThreadsN) * 4;
  // (the branch-free gather of the HS form loads for every lane, also the clamped ones of absent nodes: with fewer nodes than one
  // tile -- the 2-atom toy graphs -- the plain form is faster, 0.24 vs 0.22 ms per reverse step)
  const bool hsk = hs > 1 && q.N >= kNodes;
  if (split) {
    if (hsk) hipLaunchKernelGGL((node_post_bf16_kernel<2, f16x8, true, true>), grid, dim3(kThreadsN), sm, st, q);
    else hipLaunchKernelGGL((node_post_bf16_kernel<2, f16x8, true>), grid, dim3(kThreadsN), sm, st, q);
  } else if (hsk && !f16 && q.HP <= 64) {
    hipLaunchKernelGGL((node_post_bf16_kernel<2, bf16x8, false, true>), grid, dim3(kThreadsN), sm, st, q);
  } else if (f16) {
    if (q.HP <= 64) hipLaunchKernelGGL((node_post_bf16_kernel<2, f16x8>), grid, dim3(kThreadsN), sm, st, q);
    else hipLaunchKernelGGL((node_post_bf16_kernel<kPostMaxOB, f16x8>), grid, dim3(kThreadsN), sm, st, q);
  } else if (q.HP <= 64) hipLaunchKernelGGL(node_post_bf16_kernel<2>, grid, dim3(kThreadsN), sm, st, q);
  else hipLaunchKernelGGL(node_post_bf16_kernel<kPostMaxOB>, grid, dim3(kThreadsN), sm, st, q);
  if (hs_out) *hs_out = hs;
  if (hs > 1 && !defer_finish)
    hipLaunchKernelGGL(node_post_finish_kernel, dim3((q.N * q.H + 255) / 256), dim3(256), 0, st, q.N, q.H, hs, q.h_partial,
                       q.b2h, q.h_out);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // namespace egnn
