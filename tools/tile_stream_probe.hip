// How fast can 529 workgroups of 512 threads pull their matrix slices out of the Infinity Cache, by request shape?  Mimics the
// Krylov tile kernels' preload (wave w of a tile takes block positions w, w + 8, ...; 9 values per position and lane):
//   v8  : 8 B per lane and request, a wave instruction covers 512 contiguous bytes (the SELL layout of libgmpnp.so)
//   v16 : 16 B per lane and request (two values of a position side by side), 1 KiB per wave instruction, 10 values per position
// Reports microseconds per launch (back-to-back launches, hipEvents) and TB/s for P = 1, 2 positions per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kTiles = 529, kThreads = 512;
template <int P>
__global__ __launch_bounds__(kThreads) void v8(const double* __restrict__ a, double* out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double* base = a + (size_t)blockIdx.x * (P * 8 * 9 * 64) + lane;
  double v[P][9];
#pragma unroll
  for (int u = 0; u < P; ++u)
#pragma unroll
    for (int j = 0; j < 9; ++j) v[u][j] = base[(size_t)((w + u * 8) * 9 + j) * 64];
  double acc = 0;
#pragma unroll
  for (int u = 0; u < P; ++u)
#pragma unroll
    for (int j = 0; j < 9; ++j) acc += v[u][j];
  if (acc == 1.2345) out[0] = acc;
}
template <int P>
__global__ __launch_bounds__(kThreads) void v16(const double2* __restrict__ a, double* out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double2* base = a + (size_t)blockIdx.x * (P * 8 * 5 * 64) + lane;
  double2 v[P][5];
#pragma unroll
  for (int u = 0; u < P; ++u)
#pragma unroll
    for (int j = 0; j < 5; ++j) v[u][j] = base[(size_t)((w + u * 8) * 5 + j) * 64];
  double acc = 0;
#pragma unroll
  for (int u = 0; u < P; ++u)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc += v[u][j].x + v[u][j].y;
  if (acc == 1.2345) out[0] = acc;
}
template <class F>
static void run(const char* name, double bytes, F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  hipEventRecord(e0);
  for (int i = 0; i < 400; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 400;
  printf("%-10s %6.2f us per launch, %6.1f MB, %5.2f TB/s\n", name, us, bytes / 1e6, bytes / us / 1e6);
}
int main() {
  const size_t bytes = (size_t)64 << 20; double* a; double* out;
  hipMalloc(&a, bytes); hipMalloc(&out, 64); hipMemset(a, 0, bytes);
  run("v8  P=1", kTiles * 8.0 * 9 * 64 * 8, [&] { v8<1><<<kTiles, kThreads>>>(a, out); });
  run("v8  P=2", kTiles * 16.0 * 9 * 64 * 8, [&] { v8<2><<<kTiles, kThreads>>>(a, out); });
  run("v16 P=1", kTiles * 8.0 * 5 * 64 * 16, [&] { v16<1><<<kTiles, kThreads>>>((const double2*)a, out); });
  run("v16 P=2", kTiles * 16.0 * 5 * 64 * 16, [&] { v16<2><<<kTiles, kThreads>>>((const double2*)a, out); });
  return 0;
}
