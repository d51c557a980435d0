// Which XCD does block b of a launch run on?  s_getreg_b32 HW_REG_XCC_ID (id 20, low 4 bits) per block, for the grid shapes of the
// fused half-iteration (512-thread workgroups, 537 and 601 blocks), several launches back to back and with another kernel resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void census(int* out, int spin) {
  const unsigned x = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;   // size-1 = 3 (4 bits), offset 0, HW_REG_XCC_ID
  if (threadIdx.x == 0) out[blockIdx.x] = (int)x;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
}
__global__ void busy(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(64); }
int main() {
  int* d; hipMalloc(&d, 4096 * 4);
  hipStream_t s2; hipStreamCreate(&s2);
  for (int grid : {537, 601}) for (int mode = 0; mode < 3; ++mode) {
    int bad = 0, off0 = -1;
    for (int rep = 0; rep < 50; ++rep) {
      if (mode == 2) busy<<<300, 256, 0, s2>>>(200);
      census<<<grid, 512>>>(d, mode == 1 ? 50 : 0);
      std::vector<int> h(grid); hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
      const int off = h[0];
      if (off0 < 0) off0 = off;
      for (int b = 0; b < grid; ++b) if (h[b] != (off + b) % 8) ++bad;
    }
    printf("grid %d mode %d (0 plain, 1 long blocks, 2 beside another kernel): XCC of block 0 = %d (first launch), blocks off the round-robin deal in 50 launches: %d\n", grid, mode, off0, bad);
  }
  return 0;
}
