// Probe of the block-scaled matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 (gfx950) for the low-precision-correction
// edge kernels (precision 'f16c8'): pins, with exact small-integer data,
//   1. the A / B operand lane map for fp8 (e4m3) operands: which K index a lane's byte j holds;
//   2. what a lane's scale byte (e8m0, selected by OPSEL) multiplies: its own 32 K values, i.e. (row l & 15, block l >> 4);
//   3. v_cvt_scalef32_pk_fp8_f32: divides by the scale?  rounding, saturation with / without MODE.FP16_OVFL;
//   4. cycles per instruction and the clock the chip holds on random operands for: f16 16x16x32, scaled fp8 / fp6 K = 128,
//      and the two per-tile mixes of interest: {2 f16 + 1 scaled fp8} (f16c8) against {6 bf16} (bf16x3), one and two waves
//      per SIMD.
// standalone: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_scale_probe.hip -o tools/micro/mfma_scale_probe.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) short i16x2;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// ---- e4m3 (OCP fn) decode on the host ----
static float e4m3_to_float(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 15 && m == 7) return NAN;
  if (e == 0) v = ldexpf((float)m, -9);   // subnormal: m * 2^-3 * 2^-6
  else v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}

template <int FA, int FB>
__global__ void one_mfma(const int* areg, const int* breg, const int* sa, const int* sb, float* d, int opsel_case) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = areg[lane * 8 + i]; b[i] = breg[lane * 8 + i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (opsel_case == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FA, FB, 0, sa[lane], 0, sb[lane]);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FA, FB, 1, sa[lane], 2, sb[lane]);
  for (int i = 0; i < 4; ++i) d[lane * 4 + i] = c[i];
}

__global__ void cvt_probe(const float* in, int n, float scale, int ovfl, unsigned* out_scaled, unsigned* out_plain) {
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
  const int i = threadIdx.x;
  if (i < n) {
    i16x2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, in[2 * i], in[2 * i + 1], scale, false);
    out_scaled[i] = __builtin_bit_cast(unsigned, r);
    out_plain[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(in[2 * i], in[2 * i + 1], 0, false);
  }
}

// ---- rate kernels: operands in registers, random data, 8 row blocks x 4 column blocks of 16 x 16 per wave (the edge kernels' wave tile) ----
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
__device__ __forceinline__ unsigned long long wall() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

// MODE 0: f16 16x16x32 only (2 per tile and "chunk" of 64)        MODE 1: scaled fp8 K128 only (1 per tile)
// MODE 2: scaled fp6 K128 only (1 per tile)                        MODE 3: f16c8 mix = 2 f16 + 1 scaled fp8 per tile
// MODE 4: bf16x3 mix = 6 bf16 16x16x32 per tile                    MODE 5: 2 f16 + 1 scaled fp6      MODE 6: 2 bf16 only (bf16 reference)
template <int MODE>
__global__ __launch_bounds__(512, 2) void rate_kernel(const int* src, float* out, unsigned long long* tim, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[8][4];
  for (int r = 0; r < 8; ++r) for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // operands: A pieces for 8 row blocks (f16: 4 regs, fp8: 8 regs, both from the random source), B pieces for 4 column blocks
  f16x8 a16[8]; i32x8 a8[4]; f16x8 b16[4]; i32x8 b8[4];
  const int* s = src + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 64;
  for (int r = 0; r < 8; ++r) { int t[4]; for (int i = 0; i < 4; ++i) t[i] = s[(r * 4 + i) & 63]; a16[r] = __builtin_bit_cast(f16x8, *(int __attribute__((ext_vector_type(4)))*)t); }
  for (int c = 0; c < 4; ++c) { int t[4]; for (int i = 0; i < 4; ++i) t[i] = s[(32 + c * 4 + i) & 63]; b16[c] = __builtin_bit_cast(f16x8, *(int __attribute__((ext_vector_type(4)))*)t); }
  for (int r = 0; r < 4; ++r) for (int i = 0; i < 8; ++i) a8[r][i] = s[(r * 8 + i + 5) & 63] & 0x3F3F3F3F | (s[(r + i) & 63] & 0x80808080);
  for (int c = 0; c < 4; ++c) for (int i = 0; i < 8; ++i) b8[c][i] = s[(c * 8 + i + 11) & 63] & 0x3F3F3F3F | (s[(c + i + 3) & 63] & 0x80808080);
  const int sc = 0x7F7F7F7F;
  __syncthreads();
  const unsigned long long t0 = now(), w0 = wall();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (MODE == 0 || MODE == 3 || MODE == 5) {
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[r], b16[c], acc[r][c], 0, 0, 0);
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[(r + 1) & 7], b16[c], acc[r][c], 0, 0, 0);
        }
        if constexpr (MODE == 6 || MODE == 4) {
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a16[r]), __builtin_bit_cast(bf16x8, b16[c]), acc[r][c], 0, 0, 0);
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a16[(r + 1) & 7]), __builtin_bit_cast(bf16x8, b16[c]), acc[r][c], 0, 0, 0);
        }
        if constexpr (MODE == 4) {
          for (int k = 0; k < 4; ++k)
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a16[(r + 2 + k) & 7]), __builtin_bit_cast(bf16x8, b16[(c + k) & 3]), acc[r][c], 0, 0, 0);
        }
        if constexpr (MODE == 1 || MODE == 3)
          acc[r][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[r & 3], b8[c], acc[r][c], 0, 0, 0, sc, 0, sc);
        if constexpr (MODE == 2 || MODE == 5)
          acc[r][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[r & 3], b8[c], acc[r][c], 2, 2, 0, sc, 0, sc);
      }
  }
  const unsigned long long t1 = now(), w1 = wall();
  float v = 0.f;
  for (int r = 0; r < 8; ++r) for (int c = 0; c < 4; ++c) for (int i = 0; i < 4; ++i) v += acc[r][c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
  if (threadIdx.x == 0) { tim[2 * blockIdx.x] = t1 - t0; tim[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
static void run_rate(const char* name, int threads, const int* d_src, float* d_out, unsigned long long* d_tim, double mfma16_equiv_per_tile) {
  const int grid = 256, iters = 2000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 30; ++w) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(threads), 0, 0, d_src, d_out, d_tim, iters);   // ~1 s of warm-up: the clock settles
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(threads), 0, 0, d_src, d_out, d_tim, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::vector<unsigned long long> tim(2 * grid);
  CHECK(hipMemcpy(tim.data(), d_tim, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
  std::vector<double> cyc, clk;
  for (int i = 0; i < grid; ++i) { cyc.push_back((double)tim[2 * i]); clk.push_back((double)tim[2 * i] / (double)tim[2 * i + 1] * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  const double tiles = (double)iters * 32;          // per wave
  const double waves_per_simd = threads / 256.0;
  const double cyc_per_tile_simd = cyc[grid / 2] / tiles / waves_per_simd;   // SIMD cycles per (tile, chunk of 64 hidden units)
  const double tiles_per_s = tiles * (threads / 64) * grid / (ms * 1e-3);
  // one tile-chunk of the product a[16 x 64] W[64 x 16] = 2 * 16 * 16 * 64 FLOP of ALGORITHMIC work
  printf("%-34s %d waves/SIMD: %7.3f ms  %6.1f SIMD-cycles per tile-chunk (%.2f x the 32 of two f16 MFMAs)  held clock %.3f GHz  "
         "algorithmic %.1f TFLOP/s\n", name, (int)waves_per_simd, ms, cyc_per_tile_simd, cyc_per_tile_simd / 32.0, clk[grid / 2],
         tiles_per_s * 2.0 * 16 * 16 * 64 / 1e12);
  (void)mfma16_equiv_per_tile;
}

int main() {
  srand(7);
  // ---------- 1. lane map, e4m3 x e4m3 ----------
  // exact small integers: e4m3 encodes 0, 1, 2, 3, 4 exactly: 0x00, 0x38, 0x40, 0x44, 0x48
  const uint8_t enc[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
  int A[16][128], B[128][16];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) A[i][k] = (rand() % 9) - 4;
  for (int k = 0; k < 128; ++k) for (int j = 0; j < 16; ++j) B[k][j] = (rand() % 9) - 4;
  double ref[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 128; ++k) s += A[i][k] * B[k][j]; ref[i][j] = s; }
  auto encode = [&](int v) -> uint8_t { return (uint8_t)(enc[abs(v)] | (v < 0 ? 0x80 : 0)); };
  int *d_a, *d_b, *d_sa, *d_sb; float* d_d;
  CHECK(hipMalloc(&d_a, 64 * 8 * 4)); CHECK(hipMalloc(&d_b, 64 * 8 * 4)); CHECK(hipMalloc(&d_sa, 64 * 4)); CHECK(hipMalloc(&d_sb, 64 * 4));
  CHECK(hipMalloc(&d_d, 64 * 4 * 4));
  const char* hname[3] = {"k = 32 (l >> 4) + j                (32 contiguous K per lane)",
                          "k = 16 (l >> 4) + (j & 15) + 64 (j >> 4)   (two 16-byte halves, 64 apart)",
                          "k = 8 (l >> 4) + (j & 7) + 32 (j >> 3)     (four 8-byte pieces, 32 apart)"};
  int layout = -1;
  for (int hyp = 0; hyp < 3; ++hyp) {
    uint8_t areg[64][32], breg[64][32];
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 32; ++j) {
        const int q = l >> 4;
        const int k = hyp == 0 ? 32 * q + j : (hyp == 1 ? 16 * q + (j & 15) + 64 * (j >> 4) : 8 * q + (j & 7) + 32 * (j >> 3));
        areg[l][j] = encode(A[l & 15][k]);
        breg[l][j] = encode(B[k][l & 15]);
      }
    int sa[64], sb[64];
    for (int l = 0; l < 64; ++l) sa[l] = sb[l] = 0x7F7F7F7F;   // 2^0
    CHECK(hipMemcpy(d_a, areg, sizeof(areg), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_b, breg, sizeof(breg), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_sa, sa, sizeof(sa), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_sb, sb, sizeof(sb), hipMemcpyHostToDevice));
    hipLaunchKernelGGL((one_mfma<0, 0>), dim3(1), dim3(64), 0, 0, d_a, d_b, d_sa, d_sb, d_d, 0);
    float d[64][4];
    CHECK(hipMemcpy(d, d_d, sizeof(d), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) if (d[l][i] != (float)ref[4 * (l >> 4) + i][l & 15]) ++bad;   // C/D: col = l & 15, row = 4 (l >> 4) + i
    printf("lane map hypothesis %d  %s: %s (%d of 256 outputs differ)\n", hyp, hname[hyp], bad ? "NO" : "MATCHES", bad);
    if (!bad && layout < 0) layout = hyp;
  }
  if (layout < 0) { printf("no hypothesis matched: stop\n"); return 1; }
  // ---------- 2. scales ----------
  {
    uint8_t areg[64][32], breg[64][32];
    memset(areg, 0x38, sizeof(areg)); memset(breg, 0x38, sizeof(breg));   // all ones: D[i][j] = sum over the 4 blocks of 32 * scaleA(i, blk) * scaleB(j, blk)
    int sa[64], sb[64];
    // byte 0 of lane l: exponent (l & 15) + 4 (l >> 4) - 20, i.e. distinct per (row, block); byte 1 / byte 2 hold other values for the OPSEL case
    for (int l = 0; l < 64; ++l) {
      const int e = (l & 15) + 16 * (l >> 4);
      sa[l] = (127 + (e % 7)) | ((127 - (e % 5)) << 8) | (0x7F << 16) | (0x7F << 24);
      sb[l] = (127 - (e % 3)) | (0x7F << 8) | ((127 + (e % 4)) << 16) | (0x7F << 24);
    }
    CHECK(hipMemcpy(d_a, areg, sizeof(areg), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_b, breg, sizeof(breg), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_sa, sa, sizeof(sa), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_sb, sb, sizeof(sb), hipMemcpyHostToDevice));
    for (int oc = 0; oc < 2; ++oc) {
      hipLaunchKernelGGL((one_mfma<0, 0>), dim3(1), dim3(64), 0, 0, d_a, d_b, d_sa, d_sb, d_d, oc);
      float d[64][4];
      CHECK(hipMemcpy(d, d_d, sizeof(d), hipMemcpyDeviceToHost));
      int bad = 0;
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
          const int row = 4 * (l >> 4) + i, col = l & 15;
          double want = 0;
          for (int q = 0; q < 4; ++q) {
            const int ea = row + 16 * q, eb = col + 16 * q;
            const int xa = oc == 0 ? (ea % 7) : -(ea % 5);       // OPSEL 1 selects byte 1 of A's scale register
            const int xb = oc == 0 ? -(eb % 3) : (eb % 4);       // OPSEL 2 selects byte 2 of B's
            want += 32.0 * ldexp(1.0, xa + xb);
          }
          if (d[l][i] != (float)want) ++bad;
        }
      printf("scale semantics, opsel case %d (lane l's byte scales its own 32 K values = (row or column l & 15, block l >> 4); value 2^(byte - 127)): %s (%d differ)\n",
             oc, bad ? "NO" : "MATCHES", bad);
    }
  }
  // ---------- 3. conversions ----------
  {
    const int n = 16;
    float in[2 * n] = {0.f, 1.f, 1.0625f, 1.1875f, -3.3f, 17.f, 447.f, 449.f, 480.f, 1000.f, -1e6f, 0.0019f, 0.001f, 0.0156f, 0.3f, -0.07f,
                       1.5f, 2.5f, 3.5f, 4.5f, 5.5f, 6.5f, 7.5f, 20.f, 28.f, 0.00097f, 0.0029f, 240.f, 464.f, 465.f, 1e-8f, -0.f};
    float* d_in; unsigned *d_o1, *d_o2;
    CHECK(hipMalloc(&d_in, sizeof(in))); CHECK(hipMalloc(&d_o1, n * 4)); CHECK(hipMalloc(&d_o2, n * 4));
    CHECK(hipMemcpy(d_in, in, sizeof(in), hipMemcpyHostToDevice));
    for (int ovfl = 0; ovfl < 2; ++ovfl)
      for (int si = 0; si < 2; ++si) {
        const float scale = si == 0 ? 1.0f : 0.25f;
        hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, d_in, n, scale, ovfl, d_o1, d_o2);
        unsigned o1[n], o2[n];
        CHECK(hipMemcpy(o1, d_o1, sizeof(o1), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(o2, d_o2, sizeof(o2), hipMemcpyDeviceToHost));
        printf("cvt_scalef32_pk_fp8_f32, scale %.2f, FP16_OVFL %d:  in -> decoded (x scale)   [plain v_cvt_pk_fp8_f32 in brackets]\n", scale, ovfl);
        for (int i = 0; i < n; ++i)
          for (int h = 0; h < 2; ++h) {
            const uint8_t b1 = (o1[i] >> (8 * h)) & 0xFF, b2 = (o2[i] >> (8 * h)) & 0xFF;
            printf("   %12.6g -> 0x%02x = %10.6g   [0x%02x = %10.6g]\n", in[2 * i + h], b1, e4m3_to_float(b1) * scale, b2, e4m3_to_float(b2));
          }
      }
  }
  // ---------- 4. rates ----------
  {
    const size_t nsrc = (size_t)256 * 512 * 64;
    std::vector<int> src(nsrc);
    // random f16 bit patterns with exponents around 1 (no inf / nan): sign | exponent 13..16 | mantissa
    for (size_t i = 0; i < nsrc; ++i) {
      auto h = [&]() { return (unsigned)(((rand() & 1) << 15) | ((13 + (rand() & 3)) << 10) | (rand() & 0x3FF)); };
      src[i] = (int)(h() | (h() << 16));
    }
    int* d_src; float* d_out; unsigned long long* d_tim;
    CHECK(hipMalloc(&d_src, nsrc * 4)); CHECK(hipMalloc(&d_out, (size_t)256 * 512 * 4)); CHECK(hipMalloc(&d_tim, 2 * 256 * 8));
    CHECK(hipMemcpy(d_src, src.data(), nsrc * 4, hipMemcpyHostToDevice));
    for (int threads = 256; threads <= 512; threads += 256) {
      run_rate<6>("2 x bf16 16x16x32 (bf16 path)", threads, d_src, d_out, d_tim, 2);
      run_rate<0>("2 x f16 16x16x32 (fp16 path)", threads, d_src, d_out, d_tim, 2);
      run_rate<1>("1 x scaled e4m3 16x16x128", threads, d_src, d_out, d_tim, 2);
      run_rate<2>("1 x scaled e2m3 (fp6) 16x16x128", threads, d_src, d_out, d_tim, 1);
      run_rate<3>("f16c8: 2 x f16 + 1 x scaled e4m3", threads, d_src, d_out, d_tim, 4);
      run_rate<5>("f16c6: 2 x f16 + 1 x scaled e2m3", threads, d_src, d_out, d_tim, 3);
      run_rate<4>("bf16x3: 6 x bf16 16x16x32", threads, d_src, d_out, d_tim, 6);
    }
  }
  return 0;
}
