// Training backward, FIRST Linear layers of the two edge MLPs (EquivariantGraphNeuralNetwork.py:13-25 under autograd), factorised
// the way the forward factorises them.  The forward never forms [h_i | h_j | d2] per edge: a1[e] = P[dst e] + Q[src e] + wd d2_e
// with per-NODE tables P = W1[:, :H] h + b1, Q = W1[:, H:2H] h (node_pre).  The same linearity runs backwards: with g1 = dL/da1
// (the dgrad kernel's output, [edges, W] bf16)
//
//     dL/dP[n]   = Gd[n] = sum over the edges n RECEIVES of g1[e]            dL/dQ[n] = Gs[n] = sum over the edges n SENDS of g1[e]
//     dL/dW1[:, :H] = Gd^T h        dL/dW1[:, H:2H] = Gs^T h        dL/db1 = sum_n Gd[n]        dL/dh += Gd W1[:, :H] + Gs W1[:, H:2H]
//     dL/dW1[:, 2H] = cd = sum_e g1[e] d2_e                          dL/d(d2_e) = g1[e] . wd
//
// i.e. four segment / scatter sums over the edges and then NODE-level products (N rows instead of E = 63 N): what round 3 did with
// two weight-gradient GEMMs over all edges (gemm_tn against a gathered [E, 128] input), a row-streaming dgrad GEMM (gemm_rows),
// a gather and a scatter pass -- each reading dL/da1 again -- is ONE pass over dL/da1 here.
//
// first_reduce_kernel: workgroup = (graph, MLP, 256-column slice), see the comment at the kernel.  Graphs of more than 64 nodes
// take the round-3 chain (the caller checks).
#include "kernels.h"

namespace egnn {
namespace {

constexpr int kFRNodes = 64;     // nodes per graph the LDS accumulator of the sender sums holds
constexpr int kFRCols = 256;     // columns per workgroup

struct FirstReduceParams {
  int B, a, n;                        // graphs; edge chunk [a, a + n) of the plan's edge list (g1 rows are chunk-relative)
  const int *graph_ptr, *row_ptr, *edge_src;
  const float* x;                     // [N][3]
  const void* g1[2];                  // bf16 [n][W[mlp]]  (0 = mlp_x, 1 = mlp_m)
  int W[2];
  const float* wd[2];                 // [W] first-layer d^2 column W1[:, 2H] (unscaled)
  float *Gd[2], *Gs[2];               // [N][W] fp32, accumulated (+=)
  float* cd[2];                       // [B][W] fp32, accumulated
  float* gd2_part;                    // [nparts][n] dL/d(d2_e) shares, one per (MLP, column slice): assigned
  int nparts;
};

// Sums over the 64 lanes of 16 values per lane at once (halving butterfly: every step pairs two lanes and each keeps half of the
// value indices): 17 cross-lane moves instead of 16 x 6.  Returns, in every lane, the total of value index
// q = 8 b5 + 4 b4 + 2 b3 + b2 (b_k = bit k of the lane id).
__device__ __forceinline__ float wave_sum16(float (&v)[16], int lane) {
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int m = 32 >> st, h = 8 >> st;
    const bool up = (lane & m) != 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < h) {
        const float keep = up ? v[i + h] : v[i], send = up ? v[i] : v[i + h];
        v[i] = keep + __shfl_xor(send, m);
      }
  }
  float r = v[0];
  r += __shfl_xor(r, 2);
  r += __shfl_xor(r, 1);
  return r;
}

// LDS float atomics are NOT the way to build the sender sums: the first builds of this kernel added every element with
// ds_add_f32 (slots of rows hitting different senders) and ran 10.9 ms per call = 0.4 TB/s whatever the bank layout and whatever
// the load schedule -- ~180 cycles per 64-lane ds_add_f32.  Here every update is a plain 16-byte LDS read-modify-write and no two
// can collide: wave w owns the senders s with s % 4 == w (it walks, for every receiving node in turn, only the rows those
// senders sent: for a fully connected graph a quarter of the node's rows), one row per instruction (64 lanes x 4 columns).
// The receiving node's sum is then split over the four waves: they walk the nodes in lock step and one of them adds the four
// shares after a barrier per node.  Same sums in the same order on every run (no atomics anywhere in this kernel).
__global__ __launch_bounds__(256) void first_reduce_kernel(const FirstReduceParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_gs = reinterpret_cast<float*>(smem);                    // [kFRNodes][kFRCols] sender sums of the graph
  float* s_run = s_gs + kFRNodes * kFRCols;                        // [2][4][kFRCols] the waves' shares of a receiving node's sum
  float* s_cd = s_run + 2 * 4 * kFRCols;                           // [4][kFRCols]
  float* s_x = s_cd + 4 * kFRCols;                                 // [kFRNodes][3]
  const int g = blockIdx.x;
  const int nslice0 = p.W[0] / kFRCols;
  const int mlp = (int)blockIdx.y < nslice0 ? 0 : 1;
  const int slice = mlp == 0 ? blockIdx.y : blockIdx.y - nslice0;
  const int W = p.W[mlp];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col0 = slice * kFRCols + 4 * lane;
  const int n0 = p.graph_ptr[g], n1 = p.graph_ptr[g + 1];
  const int lo = p.a, hi = p.a + p.n;
  // nothing of this graph in the chunk: leave (graphs are contiguous in the edge list)
  if (p.row_ptr[n1] <= lo || p.row_ptr[n0] >= hi) return;
  for (int i = tid; i < (n1 - n0) * kFRCols; i += 256) s_gs[i] = 0.f;
  for (int i = tid; i < (n1 - n0) * 3; i += 256) s_x[i] = p.x[(size_t)3 * n0 + i];
  __syncthreads();
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  const __bf16* g1 = static_cast<const __bf16*>(p.g1[mlp]);
  const f32x4 wv = *reinterpret_cast<const f32x4*>(p.wd[mlp] + col0);
  f32x4 cdv = {0.f, 0.f, 0.f, 0.f};
  float* part = p.gd2_part + (size_t)blockIdx.y * p.n;
  for (int nd = n0; nd < n1; ++nd) {
    const int e0 = max(p.row_ptr[nd], lo), e1 = min(p.row_ptr[nd + 1], hi);
    const int deg = e1 - e0;
    const float xi = s_x[3 * (nd - n0)], yi = s_x[3 * (nd - n0) + 1], zi = s_x[3 * (nd - n0) + 2];
    f32x4 run = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < deg; base += 64) {
      // lane l <-> row base + l: local sender index, d^2, and whether the row is this wave's
      const int myrow = base + lane;
      int sl = 0;
      float d2l = 0.f;
      bool mine = false;
      if (myrow < deg) {
        sl = p.edge_src[e0 + myrow] - n0;
        const float dx = xi - s_x[3 * sl], dy = yi - s_x[3 * sl + 1], dz = zi - s_x[3 * sl + 2];
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);   // norm(...)**2 as the forward computes d2 (:56)
        d2l = nrm * nrm;
        mine = (sl & 3) == wave;
      }
      unsigned long long todo = __ballot(mine);
      while (todo) {   // sixteen of this wave's rows per round (a fully connected 64-node graph: the node's whole share): their
                       // loads are issued together -- the kernel is a chain of such rounds, each one memory latency long
        int rr[16];
        bf16x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          rr[u] = todo ? (int)__builtin_ctzll(todo) : -1;
          if (todo) todo &= todo - 1;
          const int rc = rr[u] >= 0 ? rr[u] : 0;
          v[u] = *reinterpret_cast<const bf16x4*>(g1 + (size_t)(e0 + base + rc - lo) * W + col0);
        }
        // sender sums: a 16-byte LDS read-modify-write per row, the read of row u + 1 issued before the add of row u (the LDS
        // round trip of every row was exposed otherwise); two consecutive rows with the SAME sender (duplicate edges) take the
        // slow path so that no update is lost
        float dots[16];
        int sj = rr[0] >= 0 ? __builtin_amdgcn_readlane(sl, rr[0]) : 0;
        f32x4 cur = *reinterpret_cast<const f32x4*>(s_gs + (size_t)sj * kFRCols + 4 * lane);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          dots[u] = 0.f;
          if (rr[u] < 0) continue;         // (wave-uniform)
          const float d2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d2l), rr[u]));
          const f32x4 f = {(float)v[u][0], (float)v[u][1], (float)v[u][2], (float)v[u][3]};
          run += f;
          cdv += f * d2;
          f32x4* gs = reinterpret_cast<f32x4*>(s_gs + (size_t)sj * kFRCols + 4 * lane);
          const int sn = (u + 1 < 16 && rr[u + 1 < 16 ? u + 1 : u] >= 0) ? __builtin_amdgcn_readlane(sl, rr[u + 1 < 16 ? u + 1 : u]) : -1;
          f32x4 nxt = cur;
          if (sn >= 0 && sn != sj) nxt = *reinterpret_cast<const f32x4*>(s_gs + (size_t)sn * kFRCols + 4 * lane);
          const f32x4 upd = cur + f;
          *gs = upd;                       // this wave alone touches sender sj
          cur = sn == sj ? upd : nxt;
          sj = sn >= 0 ? sn : sj;
          dots[u] = f[0] * wv[0] + f[1] * wv[1] + f[2] * wv[2] + f[3] * wv[3];
        }
        {
          const float tot = wave_sum16(dots, lane);
          const int qv = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
          int rq = rr[0];
#pragma unroll
          for (int u = 1; u < 16; ++u) rq = qv == u ? rr[u] : rq;
          if ((lane & 3) == 0 && rq >= 0) part[e0 + base + rq - lo] = tot;
        }
      }
    }
    // the four waves' shares of this receiving node's sum: double-buffered by node parity, one barrier per node
    float* slot = s_run + (size_t)((nd - n0) & 1) * 4 * kFRCols;
    *reinterpret_cast<f32x4*>(slot + wave * kFRCols + 4 * lane) = run;
    __syncthreads();
    if (wave == ((nd - n0) & 3) && deg > 0) {
      f32x4 tot = *reinterpret_cast<const f32x4*>(slot + 4 * lane);
#pragma unroll
      for (int w = 1; w < 4; ++w) tot += *reinterpret_cast<const f32x4*>(slot + w * kFRCols + 4 * lane);
      f32x4* gd = reinterpret_cast<f32x4*>(p.Gd[mlp] + (size_t)nd * W + col0);   // (node, columns) belongs to this thread alone
      *gd = *gd + tot;
    }
  }
  *reinterpret_cast<f32x4*>(s_cd + wave * kFRCols + 4 * lane) = cdv;
  __syncthreads();
  // sender sums of the graph and the d^2-weighted column sums: one column per thread, waves added in a fixed order
  {
    float c = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) c += s_cd[w * kFRCols + tid];
    p.cd[mlp][(size_t)g * W + slice * kFRCols + tid] += c;
    // (8 nodes per round: the loads of a round are issued together; a load / add / store chain per node is 64 serial latencies)
    float* gs = p.Gs[mlp] + (size_t)n0 * W + slice * kFRCols + tid;
    for (int b = 0; b < n1 - n0; b += 8) {
      float old[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) old[u] = b + u < n1 - n0 ? gs[(size_t)(b + u) * W] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (b + u < n1 - n0) gs[(size_t)(b + u) * W] = old[u] + s_gs[(size_t)(b + u) * kFRCols + tid];
    }
  }
}

// dL/dx from the geometry terms alone: dL/d(x_i - x_j) = g_diff[e] + 2 (sum of the dL/d(d2_e) shares + g_sq_sums[segment of i]) (x_i - x_j),
// added to g_x[i] and subtracted from g_x[j] (as egcl_backward_scatter does; the feature halves of that kernel are the node-level
// products now)
__global__ __launch_bounds__(256) void scatter_geom_kernel(int n, int nparts, const int* __restrict__ dst, const int* __restrict__ src,
                                                           const float* __restrict__ x, const float* __restrict__ gd2_part,
                                                           const float* __restrict__ g_diff, const float* __restrict__ g_S,
                                                           const int* __restrict__ node_seg, float* __restrict__ g_x) {
  // one thread per edge (round 4, first form: one thread per run of 16 edges -- a chain of 16 x (nparts + 6) dependent loads,
  // 115 us per call at 10^6 edges).  The sending side is scattered with atomics; the receiving side is summed inside the wave
  // first: the edges are sorted by receiving node, so a segmented suffix sum over the lanes (six shuffle steps) leaves each
  // node's total of this wave on the first lane of its run -- one atomic per node, wave and component.
  const int e = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const bool live = e < n;
  int i = -1, j = 0;
  float gv[3] = {0.f, 0.f, 0.f};
  if (live) {
    i = dst[e];
    j = src[e];
    float gd2 = g_S[node_seg ? node_seg[i] : 0];
    for (int s = 0; s < nparts; ++s) gd2 += gd2_part[(size_t)s * n + e];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      gv[d] = fmaf(2.0f * gd2, x[3 * i + d] - x[3 * j + d], g_diff[3 * (size_t)e + d]);
      atomicAdd(g_x + 3 * (size_t)j + d, -gv[d]);
    }
  }
  float run[3] = {gv[0], gv[1], gv[2]};
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const int io = __shfl_down(i, m);
    const bool same = lane + m < 64 && io == i;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float o = __shfl_down(run[d], m);
      if (same) run[d] += o;
    }
  }
  // (a run of equal receivers is contiguous: lane l holds the sum over lanes l .. end of its run once the doubling steps are done,
  // because every partial it took came from inside the same run)
  const int ip = __shfl_up(i, 1);
  if (live && (lane == 0 || ip != i)) {
#pragma unroll
    for (int d = 0; d < 3; ++d) atomicAdd(g_x + 3 * (size_t)i + d, run[d]);
  }
}

}  // namespace
}  // namespace egnn

using namespace egnn;

extern "C" {

int egcl_backward_first_reduce(void* stream, int B, int max_graph_nodes, int e_first, int n_edges, const int32_t* d_graph_ptr,
                               const int32_t* d_row_ptr, const int32_t* d_edge_src, const float* d_x, const void* d_g1x, int Wx,
                               const void* d_g1m, int Wm, const float* d_wdx, const float* d_wdm, float* d_Gd_x, float* d_Gs_x,
                               float* d_Gd_m, float* d_Gs_m, float* d_cd_x, float* d_cd_m, float* d_gd2_part) {
  if (B < 1 || n_edges < 1 || e_first < 0 || !d_graph_ptr || !d_row_ptr || !d_edge_src || !d_x || !d_g1x || !d_g1m || !d_wdx ||
      !d_wdm || !d_Gd_x || !d_Gs_x || !d_Gd_m || !d_Gs_m || !d_cd_x || !d_cd_m || !d_gd2_part) {
    set_error("bad egcl_backward_first_reduce arguments");
    return EGNN_EINVAL;
  }
  if (Wx % kFRCols != 0 || Wm % kFRCols != 0 || Wx < kFRCols || Wm < kFRCols || max_graph_nodes < 1 || max_graph_nodes > kFRNodes) {
    set_error("egcl_backward_first_reduce: needs hidden widths in multiples of %d and graphs of at most %d nodes (got %d, %d, %d)",
              kFRCols, kFRNodes, Wx, Wm, max_graph_nodes);
    return EGNN_EINVAL;
  }
  FirstReduceParams p;
  p.B = B; p.a = e_first; p.n = n_edges;
  p.graph_ptr = d_graph_ptr; p.row_ptr = d_row_ptr; p.edge_src = d_edge_src; p.x = d_x;
  p.g1[0] = d_g1x; p.g1[1] = d_g1m; p.W[0] = Wx; p.W[1] = Wm; p.wd[0] = d_wdx; p.wd[1] = d_wdm;
  p.Gd[0] = d_Gd_x; p.Gd[1] = d_Gd_m; p.Gs[0] = d_Gs_x; p.Gs[1] = d_Gs_m; p.cd[0] = d_cd_x; p.cd[1] = d_cd_m;
  p.gd2_part = d_gd2_part; p.nparts = (Wx + Wm) / kFRCols;
  static bool attr_done = false;
  const size_t smem = (size_t)(kFRNodes + 8 + 4) * kFRCols * sizeof(float) + (size_t)kFRNodes * 3 * sizeof(float);
  if (!attr_done) {
    EGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&first_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(first_reduce_kernel, dim3(B, p.nparts), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), p);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

int egcl_backward_scatter_geom(void* stream, int n_edges, int nparts, const int32_t* d_dst, const int32_t* d_src, const float* d_x,
                               const float* d_gd2_part, const float* d_g_diff, const float* d_g_sq_sums,
                               const int32_t* d_node_segment, float* d_g_x) {
  if (n_edges < 1 || nparts < 1 || !d_dst || !d_src || !d_x || !d_gd2_part || !d_g_diff || !d_g_sq_sums || !d_g_x) {
    set_error("bad egcl_backward_scatter_geom arguments");
    return EGNN_EINVAL;
  }
  hipLaunchKernelGGL(scatter_geom_kernel, dim3((n_edges + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n_edges, nparts,
                     d_dst, d_src, d_x, d_gd2_part, d_g_diff, d_g_sq_sums, d_node_segment, d_g_x);
  EGNN_HIP(hipGetLastError());
  return EGNN_OK;
}

}  // extern "C"
