// FETCH_SIZE calibration for the access widths used by the solver (MI355X_MICROARCH.md §HBM: the counter reads 1/2 of the
// bytes of 16-B-per-lane streams on gfx950, other widths are uncalibrated).  Streams a 2 GiB buffer (far beyond L2 +
// Infinity Cache) with (a) 16 B per lane, (b) 8 B per lane in the solver's pattern: 63 of 64 lanes active, consecutive
// wave loads 512 B apart.  Run under:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- ./calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void stream16(const double2* a, size_t n2, double* out) {
  double acc = 0; for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { double2 v = a[i]; acc += v.x + v.y; }
  if (acc == 1.2345) out[0] = acc;
}
__global__ void stream8(const double* a, size_t n, double* out) {
  double acc = 0; const int lane = threadIdx.x & 63;
  for (size_t i = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; i < n; i += (size_t)gridDim.x * 256) if (lane < 63) acc += a[i + lane];
  if (acc == 1.2345) out[0] = acc;
}
int main() {
  const size_t bytes = (size_t)2 << 30; double* a; double* out;
  hipMalloc(&a, bytes); hipMalloc(&out, 64); hipMemset(a, 0, bytes);
  for (int r = 0; r < 3; ++r) { stream16<<<4096, 256>>>((const double2*)a, bytes / 16, out); hipDeviceSynchronize(); }
  for (int r = 0; r < 3; ++r) { stream8<<<4096, 256>>>(a, bytes / 8, out); hipDeviceSynchronize(); }
  printf("bytes %zu\n", bytes); return 0;
}
