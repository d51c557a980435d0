// Feasibility probe for a peer-store transport: two PROCESSES (forked before any HIP call) on one card each allocate an uncached
// mailbox, exchange its IPC handle over pipes, map the other's, and run kernels that write the peer's mailbox (payload, system
// fence, flag) and spin on their own flag — R rounds, each checked.  Prints the time per round trip.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sys/wait.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s -> %s\n", rank, #x, hipGetErrorString(e_)); _exit(2); } } while (0)
struct Box { unsigned flag[64]; double data[2][256]; };
__global__ void exchange(Box* mine, Box* peer, unsigned seq, int rank, int n, int* err) {
  const int t = threadIdx.x;
  const int par = seq & 1;
  if (t < n) __hip_atomic_store(&peer->data[par][t], (double)(seq * 1000 + rank * 100 + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(&peer->flag[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (t == 0) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(&mine->flag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (wall_clock64() - t0 > 300000000ull) { atomicOr(err, 1); break; }   // 3 s
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  if (t < n) {
    const double v = __hip_atomic_load(&mine->data[par][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v != (double)(seq * 1000 + (1 - rank) * 100 + t)) atomicOr(err, 2);
  }
}
int main() {
  int p01[2], p10[2];
  if (pipe(p01) || pipe(p10)) return 1;
  const pid_t pid = fork();
  const int rank = pid == 0 ? 1 : 0;
  const int rd = rank == 0 ? p10[0] : p01[0], wr = rank == 0 ? p01[1] : p10[1];
  CK(hipSetDevice(0));
  Box* mine; CK(hipExtMallocWithFlags((void**)&mine, sizeof(Box), hipDeviceMallocUncached));
  CK(hipMemset(mine, 0, sizeof(Box)));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t h, hp; CK(hipIpcGetMemHandle(&h, mine));
  if (write(wr, &h, sizeof h) != (ssize_t)sizeof h) return 3;
  if (read(rd, &hp, sizeof hp) != (ssize_t)sizeof hp) return 3;
  Box* peer; CK(hipIpcOpenMemHandle((void**)&peer, hp, hipIpcMemLazyEnablePeerAccess));
  int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  char go = 1; if (write(wr, &go, 1) != 1 || read(rd, &go, 1) != 1) return 3;   // both mapped
  const int R = 2000;
  auto t0 = std::chrono::steady_clock::now();
  for (unsigned s = 1; s <= (unsigned)R; ++s) exchange<<<1, 256>>>(mine, peer, s, rank, 200, err);
  CK(hipDeviceSynchronize());
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
  int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  printf("rank %d: %d rounds, %.2f us per exchange kernel, error bits %d\n", rank, R, us, herr);
  if (write(wr, &go, 1) != 1 || read(rd, &go, 1) != 1) return 3;   // nobody unmaps while the other still runs
  CK(hipIpcCloseMemHandle(peer));
  if (rank == 0) { int st = 0; waitpid(pid, &st, 0); return herr || st ? 1 : 0; }
  return herr ? 1 : 0;
}
