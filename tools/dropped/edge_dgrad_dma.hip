// DROPPED (round 3; not built): edge_dgrad_kernel with dL/da2 staged HBM -> LDS by buffer_load ... lds three chunks ahead and a
// weight-fragment ring of eight k-steps, written to lengthen the effective lead of the HBM stream (vmcnt is in-order: a wait for
// a weight fragment also waits for every older dL/da2 request, so the register-staged kernel's lead is one chunk).  Correct
// (tests/test_training.py -m gpu green), 232 VGPRs, no spill; measured at C4 shapes, same box, interleaved: 1.925 ms against
// 1.948 ms per launch (mean of the mlp_x and mlp_m launches) = -1.2 %: the 10 % the kernel gains when dL/da2 comes from L2 is
// not latency the request distance can hide.  Kept for the record; drop into edge_bwd_dgrad.hip above init_edge_dgrad_attributes()
// to rebuild it (it uses that file's DgradParams, dgrad_epilogue, k*D constants).

// ---- the same product with dL/da2 staged by LDS-DMA (the launched form) ------------------------------------------------
// Why: `vmcnt` counts loads in issue order, so every wait for a weight fragment (requested one chunk ahead in the kernel above)
// also waits for every older dL/da2 request: the effective lead of the HBM stream was ONE chunk (~1 us) however many register
// sets were in flight (served from L2 instead of HBM the kernel ran 10 % faster).  Here the dL/da2 chunk goes HBM -> LDS by
// `buffer_load ... lds` THREE chunks ahead (no registers), which frees the registers for a weight-fragment ring of EIGHT k-steps
// = two chunks: the oldest request a fragment wait can force is then two chunks old.
//   LDS stage = [128 rows][8 pieces of 16 bytes], linear as the DMA writes it; piece j of row r holds k-group j ^ ((r >> 1) & 7)
//   (swizzle on the SOURCE side: the 8 lanes of a row still read one 128-byte line), which makes the 16 lanes of a ds_read_b128
//   phase hit 16 different 16-byte columns.  Ring of 4 stages (64 KiB per workgroup, two workgroups per CU).
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ void dg_dma16(rsrc_t rs, char* dst, unsigned voff) {   // plain function: see gemm_tn.hip (host stub quirk)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)dst, 16, voff, 0, 0, 0);
}
constexpr int kStD = 4;                       // ring stages
constexpr size_t kStageD = (size_t)kRD * 128; // bytes per stage
constexpr size_t kSmemDma = kOffA1D + kStD * kStageD;

__global__ __launch_bounds__(256, 2) void edge_dgrad_dma_kernel(const DgradParams p) {
  constexpr int CB = 2, NW = 4, BQ = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_dst = reinterpret_cast<int*>(smem + kOffDstD);
  int* s_src = reinterpret_cast<int*>(smem + kOffSrcD);
  float* s_d2 = reinterpret_cast<float*>(smem + kOffD2D);
  char* s_a1 = smem + kOffA1D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const float* wd_ptr = p.wd;
  const void* tab_ptr = p.table;
  void* out_ptr = p.g_a1_out;
  unsigned tab_bytes = (unsigned)((size_t)p.N * p.TC * 2);
  asm volatile("" : "+s"(wd_ptr), "+s"(tab_ptr), "+s"(out_ptr), "+s"(tab_bytes));
  const int nsplit = p.KP / (32 * NW * CB);
  const int j = xcd_tile(blockIdx.x, gridDim.x);
  const int tile = j / nsplit, half = j - tile * nsplit;
  const int e0 = tile * kRD;
  const int nvalid = min(kRD, p.E - e0);
  const int NC = p.Kd / kKCD, KS = p.Kd / 16;
  const rsrc_t rs_g = make_rsrc(p.g_a2, (unsigned)((size_t)p.E * p.Kd * 2));   // rows past the chunk read as zero
  const rsrc_t rs_w = make_rsrc(p.w2t, (unsigned)((size_t)p.KP * p.Kd * 2));
  // this lane's pieces of a chunk: row group 4 wave + i (8 rows each), row = 8 group + lane / 8, LDS piece lane % 8
  unsigned vsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (4 * wave + i) + (lane >> 3);
    const int kg = (lane & 7) ^ ((row >> 1) & 7);
    vsrc[i] = (unsigned)(e0 + row) * (unsigned)p.Kd * 2u + (unsigned)kg * 16u;
  }
  auto dma_chunk = [&](const int c) {   // chunk c -> stage c % 4
    char* dst = s_a1 + (size_t)(c & (kStD - 1)) * kStageD + (size_t)(4 * wave) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) dg_dma16(rs_g, dst + i * 1024, vsrc[i] + (unsigned)c * (kKCD * 2u));
  };
  // order of the first requests: edge indices, then (dependent) coordinates, THEN the three chunks -- a wait for an older
  // load does not wait for the younger DMAs, the other way round it would
  int d = 0, sn = 0;
  float cx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (tid < nvalid) {
    d = p.edge_dst[e0 + tid];
    sn = p.edge_src[e0 + tid];
#pragma unroll
    for (int k = 0; k < 3; ++k) { cx[k] = p.x[3 * d + k]; cx[3 + k] = p.x[3 * sn + k]; }
  }
  __builtin_amdgcn_sched_barrier(0);
  dma_chunk(0);
  dma_chunk(1);
  dma_chunk(2);
  __builtin_amdgcn_sched_barrier(0);
  if (tid < kRD) {
    const float dx = cx[0] - cx[3], dy = cx[1] - cx[4], dz = cx[2] - cx[5];
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);   // norm(...)**2 as in the forward (:56)
    s_dst[tid] = (int)(((unsigned)d * (unsigned)p.TC + (unsigned)p.offP) * 2u);
    s_src[tid] = (int)(((unsigned)sn * (unsigned)p.TC + (unsigned)p.offQ) * 2u);
    s_d2[tid] = tid < nvalid ? nrm * nrm : 0.f;
  }
  const unsigned lane16 = lane * 16u;
  const int colblk0 = half * NW * CB + wave * CB;
  const unsigned w0off = (unsigned)colblk0 * KS * 1024u;
  // LDS byte address of this lane's operand piece of k-step s in stage 0, row block 0: row r, k-group 2 s + hh
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)s_a1;
  unsigned afrag[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) afrag[s] = lds0 + (unsigned)r * 128u + ((unsigned)((2 * s + hh) ^ ((r >> 1) & 7)) << 4);

  f32x16 acc[kRBD][CB];
#pragma unroll
  for (int rb = 0; rb < kRBD; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;
  bf16x8 bq[BQ][CB];   // weight fragments of k-steps ks .. ks + 7 (slot = k-step % 8)
#pragma unroll
  for (int s = 0; s < BQ; ++s)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) bq[s][cb] = ldbuf_bf16x8(rs_w, lane16, w0off + ((unsigned)cb * KS + s) * 1024u);
  __builtin_amdgcn_sched_barrier(0);
#define VM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
  // raw barrier: __syncthreads() carries a fence for which hipcc waits for EVERY LDS-DMA in flight (the requests just issued
  // included); the LDS traffic of the loop is inline asm + DMA, ordered by the counted waits here
#define WG_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
  VM_WAIT(20);   // chunks 0 and 1 have landed (younger: the 4 pieces of chunk 2 and the 16 fragments)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the per-tile arrays above are in LDS
  WG_BARRIER();

  bf16x8 a[kRBD];
#define LDS_RD(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define LDS_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
  LDS_RD(a[0], afrag[0], 0); LDS_RD(a[1], afrag[0], 4096); LDS_RD(a[2], afrag[0], 8192); LDS_RD(a[3], afrag[0], 12288);

  // chunk c (par = c & 1 as a literal: the fragment ring's slots are static).  dma: request chunk c + 3; more: fragments of
  // k-steps 4 c + S + 8 exist; last: no chunk follows.
  auto chunk = [&](const int c, const int par, const bool dma, const bool more, const bool last) {
    if (dma) dma_chunk(c + 3);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned so = (unsigned)(c & (kStD - 1)) * (unsigned)kStageD, sn = (unsigned)((c + 1) & (kStD - 1)) * (unsigned)kStageD;
    const unsigned ab1 = afrag[1] + so, ab2 = afrag[2] + so, ab3 = afrag[3] + so, an0 = afrag[0] + sn;
#define GROUP(S, RB, NEXT)                                                                                    \
    {                                                                                                         \
      if (!last || (S) < 3 || (RB) == 0) LDS_WAIT(3);                                                         \
      else if ((RB) == 1) LDS_WAIT(2);                                                                        \
      else if ((RB) == 2) LDS_WAIT(1);                                                                        \
      else LDS_WAIT(0);                                                                                       \
      asm volatile("" : "+v"(a[RB]));                                                                         \
      _Pragma("unroll") for (int cb = 0; cb < CB; ++cb)                                                       \
        acc[RB][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[RB], bq[4 * par + (S)][cb], acc[RB][cb], 0, 0, 0); \
      if ((S) < 3 || !last) LDS_RD(a[RB], NEXT, (RB) * 4096);                                                 \
    }
#define KSTEP(S, NEXT)                                                                                        \
    GROUP(S, 0, NEXT) GROUP(S, 1, NEXT) GROUP(S, 2, NEXT) GROUP(S, 3, NEXT)                                   \
    if (more) {                                                                                               \
      const unsigned ksn = (unsigned)(c * 4 + (S) + BQ) * 1024u;                                              \
      _Pragma("unroll") for (int cb = 0; cb < CB; ++cb)                                                       \
        bq[4 * par + (S)][cb] = ldbuf_bf16x8(rs_w, lane16, w0off + (unsigned)cb * KS * 1024u + ksn);          \
    }
    KSTEP(0, ab1) KSTEP(1, ab2) KSTEP(2, ab3) KSTEP(3, an0)
#undef KSTEP
#undef GROUP
    __builtin_amdgcn_sched_barrier(0);
  };
  // The wait before a chunk's closing barrier publishes chunk c + 2 (read from the end of chunk c + 1 on): its pieces were
  // requested at the start of chunk c - 1, and younger than them are at most 8 fragments of chunk c - 1 and 4 pieces + 8
  // fragments of chunk c (a smaller count only waits for more).
  {
    int c = 0;
    for (; c + 2 <= NC - 4; c += 2) {            // chunks that request a chunk (c + 3 < NC) and fragments
      chunk(c, 0, true, true, false); VM_WAIT(20); WG_BARRIER();
      chunk(c + 1, 1, true, true, false); VM_WAIT(20); WG_BARRIER();
    }
    // NC is a multiple of 4 (Kd % 256 == 0): c == NC - 4 here
    chunk(c, 0, true, true, false); VM_WAIT(20); WG_BARRIER();        // requests the last chunk
    chunk(c + 1, 1, false, true, false); VM_WAIT(16); WG_BARRIER();   // chunk NC - 1 landed (younger: 2 x 8 fragments)
    chunk(c + 2, 0, false, false, false); WG_BARRIER();
    chunk(c + 3, 1, false, false, true); WG_BARRIER();
  }
#undef LDS_WAIT
#undef LDS_RD
#undef VM_WAIT
#undef WG_BARRIER

  // ---- epilogue: as above ----
  const rsrc_t rs_tab = make_rsrc(tab_ptr, tab_bytes);
  __bf16* stg = reinterpret_cast<__bf16*>(s_a1) + (size_t)wave * 32 * 72;
  __bf16* gout = static_cast<__bf16*>(out_ptr) + (size_t)e0 * p.KP + 32 * colblk0;
  dgrad_epilogue<CB>(acc, rs_tab, s_dst, s_src, s_d2, wd_ptr, stg, gout, p.KP, colblk0, nvalid, lane);
}

