// Microbenchmark 2: the real vector phase (LDS wd reads, 2 units, ds_write_b128) and a matrix phase with
// LDS A-fragment reads, alone and side by side (two waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t;
}
// mode: bit0 waves0-3 V; bit1 waves4-7 M(with LDS reads); bit2 waves4-7 V; bit3 M without LDS reads; bit4 barrier each iter
__global__ __launch_bounds__(512, 2) void k(const float* in, float* out, unsigned long long* tim, int iters, int mode) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool vwave = (wave < 4 && (mode & 1)) || (wave >= 4 && (mode & 4));
  const bool mwave = wave >= 4 && (mode & 2);
  float* s_wd = (float*)(lds + 40000);
  for (int i = tid; i < 1024; i += 512) s_wd[i] = in[i];
  f32x4 p0 = *(const f32x4*)(in + tid * 8), p1 = *(const f32x4*)(in + tid * 8 + 4);
  f32x4 q0 = *(const f32x4*)(in + tid * 8 + 16), q1 = *(const f32x4*)(in + tid * 8 + 20);
  const float d2 = in[tid];
  const int kg = tid & 7, brow = tid >> 3, r = lane & 31, hh = lane >> 5;
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  bf16x8 b0, b1;
  for (int i = 0; i < 8; ++i) { b0[i] = (__bf16)in[lane + i]; b1[i] = (__bf16)in[lane + 8 + i]; }
  __syncthreads();
  unsigned long long t0 = now();
  if (vwave) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float* wd = s_wd + (it & 15) * 64 + kg * 8;
        const f32x4 w0 = *(const f32x4*)wd, w1 = *(const f32x4*)(wd + 4);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = (__bf16)silu_f(fmaf(w0[j], d2, p0[j] + q0[j]));
          o[j + 4] = (__bf16)silu_f(fmaf(w1[j], d2, p1[j] + q1[j]));
        }
        *(bf16x8*)(lds + ((size_t)kg * 129 + brow + 64 * u) * 16 + (it & 1) * 16512) = o;
        p0[0] += (float)o[0]; q1[1] += (float)o[5];
      }
      if (mode & 16) __syncthreads();
    }
  } else if (mwave) {
    for (int it = 0; it < iters; ++it) {
      const char* cur = lds + (size_t)(it & 1) * 16512 + ((size_t)hh * 129 + r) * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 a[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          if (mode & 8) a[rb] = b0; else a[rb] = *(const bf16x8*)(cur + ((size_t)(s * 2) * 129 + 32 * rb) * 16);
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          acc[rb * 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rb], b0, acc[rb * 2], 0, 0, 0);
          acc[rb * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rb], b1, acc[rb * 2 + 1], 0, 0, 0);
        }
      }
      if (mode & 16) __syncthreads();
    }
  } else if (mode & 16) {
    for (int it = 0; it < iters; ++it) __syncthreads();
  }
  unsigned long long t1 = now();
  if (lane == 0) tim[blockIdx.x * 8 + wave] = t1 - t0;
  float keep = p0[0] + q1[1];
  for (int j = 0; j < 8; ++j) keep += acc[j][0];
  if (keep == 1.2345e-30f) out[tid] = keep;
}
int main() {
  float *in, *out; unsigned long long* tim;
  hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 20); hipMalloc(&tim, 256 * 8 * 8);
  float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 37) % 101) / 50.f - 1.f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 200;
  for (int mode : {1, 2, 10, 3, 11, 5, 19, 27, 21}) {
    k<<<256, 512, 48000>>>(in, out, tim, iters, mode); hipDeviceSynchronize();
    k<<<256, 512, 48000>>>(in, out, tim, iters, mode); hipDeviceSynchronize();
    unsigned long long t[8]; hipMemcpy(t, tim + 8 * 100, sizeof(t), hipMemcpyDeviceToHost);
    printf("mode %2d: cycles/iter  wave0=%5.0f  wave4=%5.0f\n", mode, (double)t[0] / iters, (double)t[4] / iters);
  }
  printf("legend: 1=V(w0-3) 2=M+LDS(w4-7) 10=M noLDS 3=V|M+LDS 11=V|M noLDS 5=V|V 19=V|M+LDS +barrier 27=V|M noLDS +barrier 21=V|V +barrier\n");
  return 0;
}
