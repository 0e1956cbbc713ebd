// Microbenchmark: cycles of a SiLU+pack vector phase alone, and beside a partner wave issuing MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t;
}
// mode bit0: waves 0-3 run V loop; bit1: waves 4-7 run M loop; bit2: waves 4-7 run V loop too
__global__ __launch_bounds__(512, 2) void k(const float* in, float* out, unsigned long long* tim, int iters, int mode) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool vwave = (wave < 4 && (mode & 1)) || (wave >= 4 && (mode & 4));
  const bool mwave = wave >= 4 && (mode & 2);
  f32x4 p0 = *(const f32x4*)(in + tid * 8), p1 = *(const f32x4*)(in + tid * 8 + 4);
  const float d2 = in[tid];
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[lane + i]; b[i] = (__bf16)in[lane + 8 + i]; }
  __syncthreads();
  unsigned long long t0 = now();
  if (vwave) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = (__bf16)silu_f(fmaf(p1[j], d2, p0[j] + p1[j]));
          o[j + 4] = (__bf16)silu_f(fmaf(p0[j], d2, p1[j] + p0[j]));
        }
        *(bf16x8*)(lds + ((size_t)(tid & 7) * 129 + (tid >> 3) + 64 * u) * 16) = o;
        p0[0] += (float)o[0]; p1[1] += (float)o[5];
      }
    }
  } else if (mwave) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
  }
  unsigned long long t1 = now();
  if (lane == 0) tim[blockIdx.x * 8 + wave] = t1 - t0;
  float keep = p0[0] + p1[1];
  for (int j = 0; j < 8; ++j) keep += acc[j][0];
  if (keep == 1.2345e-30f) out[tid] = keep;
}
int main() {
  float *in, *out; unsigned long long* tim;
  hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 20); hipMalloc(&tim, 256 * 8 * 8);
  float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 37) % 101) / 50.f - 1.f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 200;
  for (int mode : {1, 2, 3, 5}) {
    k<<<256, 512>>>(in, out, tim, iters, mode); hipDeviceSynchronize();
    k<<<256, 512>>>(in, out, tim, iters, mode); hipDeviceSynchronize();
    unsigned long long t[8]; hipMemcpy(t, tim + 8 * 100, sizeof(t), hipMemcpyDeviceToHost);
    printf("mode %d: per-iteration cycles: V(w0)=%.0f  w4=%.0f   [V iter = 16 SiLU + 2 ds_write_b128; M iter = 32 MFMA]\n", mode,
           (double)t[0] / iters, (double)t[4] / iters);
  }
  return 0;
}
