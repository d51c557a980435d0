// Micro-probe: cost of a device-wide barrier inside a cooperative (co-resident) launch on gfx950, with the data
// exchange pattern of the Krylov kernels (every workgroup writes a few values, the others read them after the barrier).
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe
//   ./tools/grid_barrier_probe [workgroups] [threads] [barriers]
// Every spin loop is bounded by a wall-clock budget (s_memrealtime): a barrier that cannot complete sets a flag and
// every wave leaves, so the grid always drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Barrier { unsigned count; unsigned gen; unsigned dead; unsigned pad; };

__device__ inline bool grid_barrier(Barrier* b, unsigned nwg, unsigned& gen_local, unsigned long long budget) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();  // release: this workgroup's writes are visible device-wide
    const unsigned target = gen_local + 1;
    if (__hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&b->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) {
        if (__hip_atomic_load(&b->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
        if (wall_clock64() - t0 > budget) { __hip_atomic_store(&b->dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __threadfence();  // acquire
  }
  gen_local++;
  // broadcast ok through LDS
  __shared__ int okflag;
  if (threadIdx.x == 0) okflag = ok;
  __syncthreads();
  return okflag != 0;
}

__global__ __launch_bounds__(512) void k_probe(Barrier* b, double* buf, int nbar, unsigned long long budget, int* mism, int mode) {
  const unsigned nwg = gridDim.x;
  unsigned gen = 0;
  const int wg = blockIdx.x, t = threadIdx.x;
  int bad = 0;
  for (int it = 0; it < nbar; ++it) {
    double* cur = buf + (size_t)(it & 1) * nwg * 64;
    if (mode >= 1 && t < 64) cur[(size_t)wg * 64 + t] = (double)(it * 1000003 + wg * 64 + t);
    if (!grid_barrier(b, nwg, gen, budget)) return;
    if (mode >= 1 && t < 64) {
      const int src = (wg * 37 + 11 + it) % nwg;
      const double v = cur[(size_t)src * 64 + t];
      if (v != (double)(it * 1000003 + src * 64 + t)) bad++;
    }
  }
  if (bad) atomicAdd(mism, bad);
}

int main(int argc, char** argv) {
  int nwg = argc > 1 ? atoi(argv[1]) : 600, nt = argc > 2 ? atoi(argv[2]) : 512, nbar = argc > 3 ? atoi(argv[3]) : 1000;
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
  int per_cu = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_probe, nt, 0));
  printf("CUs %d, cooperative %d, blocks/CU %d -> capacity %d, clock rate %d kHz\n", prop.multiProcessorCount, prop.cooperativeLaunch, per_cu,
         per_cu * prop.multiProcessorCount, prop.clockRate);
  Barrier* b; double* buf; int* mism;
  CHECK(hipMalloc(&b, sizeof(Barrier))); CHECK(hipMalloc(&buf, sizeof(double) * 2 * nwg * 64)); CHECK(hipMalloc(&mism, 4));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  unsigned long long budget = 100000000ull;  // 1 s of the 100 MHz constant clock
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemsetAsync(b, 0, sizeof(Barrier), st)); CHECK(hipMemsetAsync(mism, 0, 4, st));
      void* args[] = {&b, &buf, &nbar, &budget, &mism, &mode};
      CHECK(hipEventRecord(e0, st));
      CHECK(hipLaunchCooperativeKernel((const void*)k_probe, dim3(nwg), dim3(nt), args, 0, st));
      CHECK(hipEventRecord(e1, st));
      CHECK(hipStreamSynchronize(st));
      float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
      Barrier hb; int hm; CHECK(hipMemcpy(&hb, b, sizeof(hb), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
      printf("mode %d rep %d: %d wg x %d thr, %d barriers: %.3f ms -> %.3f us/barrier, dead %u, mismatches %d\n", mode, rep, nwg, nt, nbar, ms,
             1e3 * ms / nbar, hb.dead, hm);
    }
  }
  return 0;
}
