// "measure the one-launch layer instead of arguing it".  What a reverse step of ONE 20-64-atom graph
// (parts/train_per_iretation.py:335-364) costs as 14 dependent launches replayed from a hipGraph, against the same phases inside
// ONE persistent launch whose workgroups all sit on ONE XCD (the graph's tiles fit 32 CUs, so no cross-XCD coherence is needed)
// and meet at an XCD-local barrier.  The phases are STAND-INS of the real kernels: same number of workgroups, same duration
// (spun on s_memrealtime, so the duration does not depend on the clock), and a real hand-off -- every workgroup publishes a 1 KiB
// record per phase and checks a neighbour's record of the previous phase behind the barrier (a stale read is counted).
//   per layer: node_pre 7.0 us (2 workgroups), coordinate + message kernel side by side 21.5 us (126 + 63 workgroups of the small
//   tiles: here 32 workgroups x 2 rounds, i.e. the persistent form takes 2 x 21.5 us unless the tiles are halved), node_post 20.6 us
//   (8 workgroups); + the fused update 6 us  (profiles/r04h_latency.log: 208 us of kernel time, 0.261 ms per replayed step)
// Three forms:
//   A  hipGraph of the 13 dependent kernels per step (X and M forked), replayed: what the library does today
//   B  one persistent launch per step, XCD-local barrier: producer stores + s_waitcnt vmcnt(0) + relaxed agent atomic arrive,
//      consumer polls with sc1 loads and reads the records with sc1 loads (same XCD: L2 is coherent, only the reader's L1 must be
//      bypassed; no release / acquire fence, MI355X_MICROARCH.md "Valid forms")
//   C  as B with agent-scope release / acquire fences (what a cross-XCD hand-off would need)
// and the bare cost of one barrier in B and C.
// standalone: hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_barrier_probe.hip -o tools/micro/xcd_barrier_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ unsigned long long wall() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }   // 100 MHz
__device__ __forceinline__ void spin_ticks(unsigned ticks) {
  const unsigned long long t0 = wall();
  while (wall() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

// ---- form A: one stand-in kernel per phase ----
__global__ void phase_kernel(unsigned ticks, unsigned* rec, int phase_id) {
  spin_ticks(ticks);
  if (threadIdx.x < 256) rec[(size_t)blockIdx.x * 256 + threadIdx.x] = (unsigned)phase_id * 1000u + threadIdx.x;
}

// ---- forms B / C: persistent launch on one XCD ----
struct Phase { unsigned ticks; int wgs; };
constexpr int kMaxPhases = 16;
struct PersistentParams {
  Phase ph[kMaxPhases];
  int nphases, steps, nw, fences, target_xcc;
  unsigned* rec;        // [nw][256]
  unsigned* counters;   // [0]: arrival tickets of the target XCD, [32]: barrier counter, [64]: stale reads, [65]: timeouts
};

__global__ __launch_bounds__(256) void persistent_kernel(const PersistentParams p) {
  const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;   // HW_REG_XCC_ID
  if (xcc != p.target_xcc) return;
  __shared__ int s_idx;
  if (threadIdx.x == 0) s_idx = (int)__hip_atomic_fetch_add(&p.counters[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int idx = s_idx;
  if (idx >= p.nw) return;   // surplus workgroups of this XCD leave before any barrier
  unsigned* bar = p.counters + 32;
  unsigned epoch = 0;
  for (int st = 0; st < p.steps; ++st) {
    for (int k = 0; k < p.nphases; ++k) {
      const unsigned tag = (unsigned)(st * p.nphases + k + 1) * 1000u;
      if (idx < p.ph[k].wgs) spin_ticks(p.ph[k].ticks);
      p.rec[((size_t)((st * p.nphases + k) & 1) * 64 + idx) * 256 + threadIdx.x] = tag + threadIdx.x;   // this phase's output record (two record sets: a
                                                                                      // workgroup that sits a phase out may already write the next one)
      // ---- barrier ----
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      ++epoch;
      if (threadIdx.x == 0) {
        if (p.fences) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = epoch * (unsigned)p.nw;
        const unsigned long long t0 = wall();
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          __builtin_amdgcn_s_sleep(1);
          if (wall() - t0 > 100000000ull) { __hip_atomic_fetch_add(&p.counters[65], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // 1 s
        }
        if (p.fences) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      }
      __syncthreads();
      // ---- the next phase's input: a neighbour's record of this phase (sc1 load: served by the XCD's L2, not this CU's L1) ----
      const int nb = (idx + 1) % p.nw;
      const unsigned got = __hip_atomic_load(&p.rec[((size_t)((st * p.nphases + k) & 1) * 64 + nb) * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (got != tag + threadIdx.x) __hip_atomic_fetch_add(&p.counters[64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();   // (a record set is rewritten two phases later, i.e. behind the NEXT barrier, which the reader has passed)
    }
  }
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  // phases of one reverse step (us, workgroups), 4 layers + update; X and M of a layer side by side
  struct Ph { double us; int wgs; const char* name; };
  const Ph layer[3] = {{7.0, 2, "node_pre"}, {21.5, 32, "edge X + M (32 workgroups per round)"}, {20.6, 8, "node_post"}};
  std::vector<Ph> phases;
  for (int l = 0; l < 4; ++l) for (int k = 0; k < 3; ++k) phases.push_back(layer[k]);
  phases.push_back({6.0, 1, "update"});
  double sum_us = 0;
  for (auto& q : phases) sum_us += q.us;
  printf("stand-in phases per step: %zu, sum of durations %.1f us\n", phases.size(), sum_us);

  unsigned *d_rec, *d_cnt;
  CHECK(hipMalloc(&d_rec, 1024 * 256 * 4)); CHECK(hipMalloc(&d_cnt, 128 * 4));
  hipStream_t st, side;
  CHECK(hipStreamCreate(&st)); CHECK(hipStreamCreate(&side));
  hipEvent_t e0, e1, ef, ej;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));

  // ---------- form A: hipGraph of 8 steps x 17 launches (X and M forked as in the library) ----------
  {
    const int steps_per_graph = 8;
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int s = 0; s < steps_per_graph; ++s)
      for (size_t k = 0; k < phases.size(); ++k) {
        const unsigned ticks = (unsigned)(phases[k].us * 100);
        if (phases[k].wgs == 32) {   // the edge pass: coordinate kernel (126 workgroups) and message kernel (63) on two streams
          CHECK(hipEventRecord(ef, st)); CHECK(hipStreamWaitEvent(side, ef, 0));
          hipLaunchKernelGGL(phase_kernel, dim3(126), dim3(512), 0, st, ticks, d_rec, (int)k);
          hipLaunchKernelGGL(phase_kernel, dim3(63), dim3(512), 0, side, (unsigned)(17.6 * 100), d_rec + 126 * 256, (int)k);
          CHECK(hipEventRecord(ej, side)); CHECK(hipStreamWaitEvent(st, ej, 0));
        } else {
          hipLaunchKernelGGL(phase_kernel, dim3(phases[k].wgs), dim3(256), 0, st, ticks, d_rec, (int)k);
        }
      }
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 5; ++w) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    std::vector<double> t;
    for (int r = 0; r < 9; ++r) {
      CHECK(hipEventRecord(e0, st));
      for (int w = 0; w < 4; ++w) CHECK(hipGraphLaunch(ge, st));
      CHECK(hipEventRecord(e1, st));
      CHECK(hipStreamSynchronize(st));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      t.push_back(ms * 1000.0 / (4 * steps_per_graph));
    }
    printf("A  hipGraph replay, 17 launches per step (13 dependent links):   %.1f us per step  = sum of durations + %.1f us (%.2f us per dependent link)\n",
           median(t), median(t) - sum_us, (median(t) - sum_us) / 13.0);
  }
  // ---------- forms B / C ----------
  for (int fences = 0; fences < 2; ++fences)
    for (int mode = 0; mode < 2; ++mode) {   // mode 0: barriers only (zero-length phases), 1: the step's phases
      PersistentParams p{};
      p.nw = 32; p.fences = fences; p.rec = d_rec; p.counters = d_cnt; p.steps = 16;
      p.nphases = (int)phases.size();
      for (int k = 0; k < p.nphases; ++k) {
        p.ph[k].wgs = std::min(phases[k].wgs, p.nw);
        // the edge phase: 189 small-tile workgroups on 32 CUs of ONE XCD = 6 rounds instead of 1 on 256 CUs -- the price of staying on one XCD
        p.ph[k].ticks = mode == 0 ? 0u : (unsigned)(phases[k].us * 100);
      }
      std::vector<double> t;
      unsigned hc[128];
      for (int r = 0; r < 7; ++r) {
        for (int xcc_try = 0; xcc_try < 1; ++xcc_try) {
          p.target_xcc = r % 8;
          CHECK(hipMemsetAsync(d_cnt, 0, 128 * 4, st));
          CHECK(hipEventRecord(e0, st));
          hipLaunchKernelGGL(persistent_kernel, dim3(8 * 48), dim3(256), 0, st, p);   // 48 per XCD dealt round-robin: 32 take part
          CHECK(hipEventRecord(e1, st));
          CHECK(hipStreamSynchronize(st));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          CHECK(hipMemcpy(hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost));
          if (hc[65] || hc[0] < (unsigned)p.nw) printf("   (run %d: %u workgroups arrived on XCD %d, %u barrier timeouts)\n", r, hc[0], p.target_xcc, hc[65]);
          t.push_back(ms * 1000.0 / p.steps);
        }
      }
      const double per_step = median(t);
      if (mode == 0)
        printf("%s  persistent launch on one XCD (32 workgroups), %s: %.2f us per barrier (%d barriers per step, stale reads %u)\n",
               fences ? "C" : "B", fences ? "agent release / acquire fences" : "sc1 loads, no fence", per_step / p.nphases, p.nphases, hc[64]);
      else
        printf("%s  ... with the step's phases: %.1f us per step = sum of durations + %.1f us (stale reads %u)\n", fences ? "C" : "B",
               per_step, per_step - sum_us, hc[64]);
    }
  printf("note: on ONE XCD the edge phase of a 64-atom graph (126 + 63 small-tile workgroups) is 6 rounds of 32 CUs instead of one round of 256:\n"
         "      + 5 x 21.5 us per layer unless the tiles grow; the persistent form can only win for graphs whose tiles fit 32 CUs in one round (<= 20 atoms).\n");
  return 0;
}
