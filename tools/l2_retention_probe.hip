// Does an XCD's L2 keep lines across a kernel boundary?  Kernel W: block b writes chunk b (plain stores).  Kernel R (next launch
// in the same stream): block b reads chunk (b + shift): shift 0 = the chunk the SAME block index wrote (same XCD under the
// round-robin deal), shift 1 = a chunk written on the neighbouring XCD, shift 8 = another block's chunk on the same XCD.
// Reports R's time per launch for working sets of 2 MB and 16 MB (4 MB of L2 per XCD, 32 MB in all).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void kw(double* a, int per_block, double v) {
  double* p = a + (size_t)blockIdx.x * per_block;
  for (int i = threadIdx.x; i < per_block; i += 256) p[i] = v + i;
}
__global__ __launch_bounds__(256) void kr(const double* a, int per_block, int shift, double* out) {
  const int b = (blockIdx.x + shift) % gridDim.x;
  const double* p = a + (size_t)b * per_block;
  double acc = 0;
  for (int i = threadIdx.x; i < per_block; i += 256) acc += p[i];
  if (acc == 1.2345) out[0] = acc;
}
int main() {
  double* a; double* out; hipMalloc(&a, 64 << 20); hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mb : {2, 16}) {
    const int blocks = 1024, per_block = (mb << 20) / 8 / blocks;
    for (int shift : {0, 1, 8}) {
      float tot = 0;
      for (int rep = 0; rep < 220; ++rep) {
        kw<<<blocks, 256>>>(a, per_block, (double)rep);
        hipEventRecord(e0);
        kr<<<blocks, 256>>>(a, per_block, shift, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 20) tot += ms;
      }
      printf("%2d MB, reader shift %d: %.2f us per read launch\n", mb, shift, tot * 1e3 / 200);
    }
  }
  return 0;
}
