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
  const float* table;  // [N][TC]
  int TC, WxP, WmP, MP, cbx, cbm;
  const float *wdx, *wdm, *b2x, *w3x, *b2m, *wa, *scal;
  const void *w2x, *w2m;
  float *agg_m, *agg_x, *part_m, *part_x;
  int dbg;  // timing experiments only (EGNN_DEBUG): bit0 drop weight loads, bit1 drop table loads
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


int launch_edge_bf16_v2(const EdgeParams& p, int tiles, hipStream_t st);
bool edge_bf16_v2_supported(const EdgeParams& p);

}  // namespace egnn
