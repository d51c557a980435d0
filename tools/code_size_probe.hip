// Micro-probe: how much of a small kernel's duration on gfx950 is instruction fetch?  Kernels whose only difference
// is N bytes of straight-line (executed once) ALU code in front of one load + one store, timed back to back.
//   hipcc --offload-arch=gfx950 -O3 tools/code_size_probe.hip -o tools/code_size_probe && ./tools/code_size_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N>
__global__ __launch_bounds__(512) void k_pad(const double* __restrict__ in, double* __restrict__ out) {
  double v = in[threadIdx.x];
  int a = threadIdx.x;
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 3" : "+v"(a));  // 2 x 8-byte VOP... per step
  out[blockIdx.x * 512 + threadIdx.x] = v + a;
}

template <int N>
float run(const double* in, double* out, int wgs, int reps, hipStream_t st) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_pad<N>, dim3(wgs), dim3(512), 0, st, in, out);
  CHECK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_pad<N>, dim3(wgs), dim3(512), 0, st, in, out);
  CHECK(hipEventRecord(e1, st));
  CHECK(hipStreamSynchronize(st));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return 1e3f * ms / reps;
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 15;
  double *in, *out; CHECK(hipMalloc(&in, 512 * 8)); CHECK(hipMalloc(&out, (size_t)wgs * 512 * 8)); CHECK(hipMemset(in, 0, 512 * 8));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  const int reps = 2000;
  printf("%d workgroups x 512 threads, back-to-back launches\n", wgs);
  printf("pad    0 steps: %.2f us\n", run<0>(in, out, wgs, reps, st));
  printf("pad   64 steps (~0.5-1 KB): %.2f us\n", run<64>(in, out, wgs, reps, st));
  printf("pad  256 steps (~2-4 KB): %.2f us\n", run<256>(in, out, wgs, reps, st));
  printf("pad  512 steps (~4-8 KB): %.2f us\n", run<512>(in, out, wgs, reps, st));
  printf("pad 1024 steps (~8-16 KB): %.2f us\n", run<1024>(in, out, wgs, reps, st));
  printf("pad 2048 steps (~16-32 KB): %.2f us\n", run<2048>(in, out, wgs, reps, st));
  return 0;
}
