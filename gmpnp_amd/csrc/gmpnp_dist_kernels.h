// Kernels of the mesh-partitioned solve (one handle per rank): packing of the ghost rows that travel after
// each BiCGStab half-iteration, the per-rank sums that travel in the one all-reduce of a half-iteration, and the device side
// of the in-process rehearsal transport.  The tile and coarse kernels themselves are those of the single-GPU solver
// (gmpnp_kernels.h): they run the owned tiles only (Ctx::tile0) and, with Ctx::dist, read all-reduced sums.
#pragma once
#include "gmpnp_kernels.h"

namespace gmpnp {

struct VecList { const double* p[4]; };
struct VecListW { double* p[4]; };

// buf[((k * nvec) + v) * width + f] = src_v[nodes[k] * width + f]: node-major, so the rows for one neighbour (a contiguous
// range of k) are one contiguous message whatever the number of vectors.
__device__ __forceinline__ void halo_pack_entry(const VecList& src, int nvec, int width, const int32_t* __restrict__ nodes, int n_nodes,
                                                double* __restrict__ buf, int i) {
  if (i >= n_nodes * nvec * width) return;
  const int f = i % width, v = (i / width) % nvec, k = i / (width * nvec);
  buf[i] = src.p[v][(size_t)nodes[k] * width + f];
}
__device__ __forceinline__ void halo_unpack_entry(const VecListW& dst, int nvec, int width, const int32_t* __restrict__ nodes, int n_nodes,
                                                  const double* __restrict__ buf, int i) {
  if (i >= n_nodes * nvec * width) return;
  const int f = i % width, v = (i / width) % nvec, k = i / (width * nvec);
  dst.p[v][(size_t)nodes[k] * width + f] = buf[i];
}
__global__ __launch_bounds__(256) void k_halo_pack(const VecList src, int nvec, int width, const int32_t* __restrict__ nodes, int n_nodes,
                                                    double* __restrict__ buf) {
  halo_pack_entry(src, nvec, width, nodes, n_nodes, buf, blockIdx.x * 256 + threadIdx.x);
}
__global__ __launch_bounds__(256) void k_halo_unpack(const VecListW dst, int nvec, int width, const int32_t* __restrict__ nodes, int n_nodes,
                                                      const double* __restrict__ buf) {
  halo_unpack_entry(dst, nvec, width, nodes, n_nodes, buf, blockIdx.x * 256 + threadIdx.x);
}

// One WORKGROUP per output value: a fixed-order sum of per-tile scalar partials or of per-slot restriction partials (a
// twice-refined mesh has 28,000 tiles and 3,500 slots per aggregate: one wave per output took 179 us per launch there).
//   phase 0 (start of a solve)  out[d]                = sum_slot cpart_v[1][slot][d]                 (P^T r_0, left by k_krylov_init)
//   phase 1 (after half A)      out[0..1]             = sum_tile part_a, part_rr ; out[2 + w n + d] = sum_slot (v, r, p)[par]
//   phase 2 (after half B)      out[0..3]             = sum_tile part_b[m]       ; out[4 + d]       = sum_slot cpart_t
//   phase 3 (end of a solve)    out[d]                = sum_slot cpart_v[0][slot][d]                 (P^T y, left by k_restrict)
// WT: the result leaves as an agent-scope write-through store (the exchange kernel: another workgroup, on any XCD, reads it
// back in the same launch and later overwrites it — no address may be dirty in two L2s); else a plain store.
// where output o of a phase comes from: `count` partials, `stride` doubles apart
__device__ __forceinline__ const double* dist_reduce_source(const Ctx& c, int phase, int par, int o, int& count, int& stride) {
  const int n = c.ncoarse;
  const int nscal = phase == 1 ? 2 : (phase == 2 ? 4 : 0);
  if (o < nscal) {
    count = c.ntiles; stride = 1;
    return phase == 1 ? (o == 0 ? c.part_a : c.part_rr) : c.part_b + (size_t)o * c.ntiles;
  }
  const int q = o - nscal, w = q / n, d = q - w * n;
  const int nf = n / c.nagg, ag = d / nf;   // part[aggregate][slot][field] (cpart_index)
  count = c.tile_slots; stride = nf;
  return (phase == 0 ? c.cpart_v[1] : phase == 3 ? c.cpart_v[0] : phase == 2 ? c.cpart_t
          : (w == 0 ? c.cpart_v[par] : w == 1 ? c.cpart_r[par] : c.cpart_p[par])) + (size_t)ag * c.tile_slots * nf + (d - ag * nf);
}
template <bool WT = false>
__device__ __forceinline__ void dist_reduce_block(const Ctx& c, int phase, int par, double* __restrict__ out, int o) {
  __shared__ double lds[4];
  const int t = threadIdx.x;
  int count, stride;
  const double* p = dist_reduce_source(c, phase, par, o, count, stride);
  double v[1] = {0.0};
  for (int i0 = t; i0 < count; i0 += 8 * 256) {   // eight independent requests per thread and trip
    double w8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w8[u] = p[(size_t)min(i0 + u * 256, count - 1) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[0] += (i0 + u * 256 < count) ? w8[u] : 0.0;
  }
  block_sum<1>(v, lds);
  if (t == 0) {
    if (WT) __hip_atomic_store(out + o, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else out[o] = v[0];
  }
}
__global__ __launch_bounds__(256) void k_dist_reduce(const Ctx c, int phase, int par, double* __restrict__ out) {
  dist_reduce_block(c, phase, par, out, blockIdx.x);
}
// The per-rank sums of a half-iteration AND the packing of the ghost rows that follow it, in one launch (partitioned solve):
// workgroups [0, nout) reduce, the others pack.
__global__ __launch_bounds__(256) void k_dist_reduce_pack(const Ctx c, int phase, int par, double* __restrict__ out, int nout, const VecList src, int nvec,
                                                           const int32_t* __restrict__ nodes, int n_nodes, double* __restrict__ buf, int width) {
  if ((int)blockIdx.x < nout) dist_reduce_block(c, phase, par, out, blockIdx.x);
  else halo_pack_entry(src, nvec, width, nodes, n_nodes, buf, ((int)blockIdx.x - nout) * 256 + (int)threadIdx.x);
}
// Coarse kernel of a half-iteration with the unpacking of the ghost rows received before it riding along: workgroups
// [0, nagg) are the coarse workgroups, the others scatter the receive buffer (the tile kernel is the next launch).
template <int NF>
__global__ __launch_bounds__(kCoarseThreads) void k_coarse_a_unpack(const Ctx c, const int k, const VecListW dst, int nvec, const int32_t* __restrict__ nodes,
                                                                    int n_nodes, const double* __restrict__ buf) {
  if ((int)blockIdx.x < c.nagg) coarse_a_body<NF, false>(c, k, blockIdx.x, 0u);
  else halo_unpack_entry(dst, nvec, NF, nodes, n_nodes, buf, ((int)blockIdx.x - c.nagg) * kCoarseThreads + (int)threadIdx.x);
}
template <int NF>
__global__ __launch_bounds__(kCoarseThreads) void k_coarse_b_unpack(const Ctx c, const int k, const VecListW dst, int nvec, const int32_t* __restrict__ nodes,
                                                                    int n_nodes, const double* __restrict__ buf) {
  if ((int)blockIdx.x < c.nagg) coarse_b_body<NF, false>(c, k, blockIdx.x, 0u);
  else halo_unpack_entry(dst, nvec, NF, nodes, n_nodes, buf, ((int)blockIdx.x - c.nagg) * kCoarseThreads + (int)threadIdx.x);
}

// ||b||^2 of the owned rows (k_res_gather's per-workgroup partials) and the four status bits, as doubles for the all-reduce
__global__ __launch_bounds__(256) void k_norm_reduce(const double* __restrict__ part, int nblocks, const int32_t* __restrict__ status,
                                                      double* __restrict__ out) {
  __shared__ double lds[4];
  double v[1] = {0.0};
  for (int i = threadIdx.x; i < nblocks; i += 256) v[0] += part[i];
  block_sum<1>(v, lds);
  if (threadIdx.x == 0) {
    out[0] = v[0];
    const int st = *status;
    for (int b = 0; b < 4; ++b) out[1 + b] = (st >> b) & 1 ? 1.0 : 0.0;
  }
}

// three dot-product partial arrays (k_dots3 layout: part[m * nblocks + block]) -> out[0..2]
__global__ __launch_bounds__(256) void k_dots3_reduce(const double* __restrict__ part, int nblocks, double* __restrict__ out) {
  __shared__ double lds[12];
  double v[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nblocks; i += 256) { v[0] += part[i]; v[1] += part[nblocks + i]; v[2] += part[2 * nblocks + i]; }
  block_sum<3>(v, lds);
  if (threadIdx.x == 0) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; }
}

// Galerkin matrix before its all-reduce: rows of aggregates this rank does not own come from ghost (identity) rows
__global__ __launch_bounds__(256) void k_zero_foreign_rows(double* __restrict__ Ac, int n, int row0, int row1) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n * n) return;
  const int r = q / n;
  if (r < row0 || r >= row1) Ac[q] = 0.0;
}

// ---- peer-mailbox transport -------------------------------------------------------------------------------------------------
// One process per rank.  Every rank owns a MAILBOX in its GPU's memory (uncached allocation, mapped into every other rank's
// process through an IPC handle; over xGMI when the ranks sit on different GPUs).  An exchange is ONE launch of one
// workgroup per rank: it stores its contribution to an all-reduce into every rank's mailbox and its ghost-row messages into
// the neighbours' (system-scope stores), fences, raises its flag in every mailbox to the exchange's sequence number, waits
// until every rank's flag in its OWN mailbox has reached that number, sums the contributions in rank order (the same order,
// hence the same bits, on every rank) and copies the received rows into the handle's receive buffer.  The flag store is a
// system-scope release, the end of the wait a system-scope acquire fence (one workgroup: one of each per launch).  No collective library,
// no host step; a half-iteration of the partitioned BiCGStab costs one such launch instead of an all-reduce and a grouped
// send/receive.  Slots alternate with the parity of the sequence number: a rank can be at most one exchange ahead of the
// slowest (it needs everybody's flag of exchange k to finish k, and a rank raises k only after it has consumed k - 1).
// Mailbox layout (bytes): [flags: kPeerMax x 128] [recv-offset table: kPeerMax x int32 at 1024] [red at red_off: 2 x size x
// red_cap doubles] [halo at halo_off: per neighbour segment j both parities side by side, (2 recv_ptr[j] + parity len_j) wmax].
constexpr int kPeerMax = kXRanksMax;       // ranks of a partition
constexpr int kPeerNbMax = 8;     // neighbours of one rank
constexpr int kPeerFlagStride = kXFlagStride;  // uint32 words between two flags (one 128-B line each)
constexpr size_t kPeerTableOff = 1024, kPeerRedOff = 2048;
constexpr int kLLHaloNodes = 2048;   // ghost nodes the flagged-word area of a mailbox holds (3.5 MB); a rank with more uses the flag-based launches
struct PeerArgs {
  unsigned char* box[kPeerMax];   // every rank's mailbox as mapped in THIS process (box[me]: the own allocation)
  int me, size;
  unsigned seq;                   // number of this exchange: the same on every rank, strictly increasing
  size_t halo_off;                // byte offset of the ghost-row area (the same on every rank: red_cap is)
  int red_cap, wmax;              // doubles per contribution slot; widest ghost row (doubles per node) = unit of the halo area
  int n_nb, nb_rank[kPeerNbMax], send_ptr[kPeerNbMax + 1], recv_ptr[kPeerNbMax + 1];   // node offsets, as in the handle
  int peer_recv_ptr[kPeerNbMax];  // recv_ptr of THIS rank's segment in neighbour j's plan
  int32_t* err;                   // pinned host word (bit 0: a flag did not arrive within the budget)
  unsigned long long budget;      // wall_clock64 ticks a flag may take (5 s at the device's hipDeviceAttributeWallClockRate)
};
__device__ __forceinline__ void st_sys(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double ld_sys(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// red_src[0..n_red) summed over the ranks into red_dst (n_red = 0: no all-reduce); `per` doubles per ghost node from sendbuf to the
// neighbours and from them into recvbuf (per = 0: no rows)
__global__ __launch_bounds__(1024) void k_peer_exchange(const PeerArgs a, const double* __restrict__ red_src, int n_red, double* __restrict__ red_dst,
                                                        const double* __restrict__ sendbuf, double* __restrict__ recvbuf, int per) {
  const int t = threadIdx.x, nt = blockDim.x, par = a.seq & 1;
  if (n_red > 0)
    for (int q = 0; q < a.size; ++q) {
      double* dst = reinterpret_cast<double*>(a.box[q] + kPeerRedOff) + ((size_t)par * a.size + a.me) * a.red_cap;
      for (int i = t; i < n_red; i += nt) st_sys(dst + i, red_src[i]);
    }
  if (per > 0)
    for (int j = 0; j < a.n_nb; ++j) {
      const int len = a.send_ptr[j + 1] - a.send_ptr[j], n = len * per;
      const double* src = sendbuf + (size_t)a.send_ptr[j] * per;
      double* dst = reinterpret_cast<double*>(a.box[a.nb_rank[j]] + a.halo_off) + ((size_t)2 * a.peer_recv_ptr[j] + (size_t)par * len) * a.wmax;
      for (int i = t; i < n; i += nt) st_sys(dst + i, src[i]);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every thread: its write-through stores are out before the workgroup's flags
  __syncthreads();
  // release / acquire at system scope, ONCE per launch (this is the only workgroup): the flag is what a peer GPU synchronises on
  if (t < a.size)
    __hip_atomic_store(reinterpret_cast<unsigned*>(a.box[t]) + a.me * kPeerFlagStride, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (t < a.size) {
    const unsigned* f = reinterpret_cast<const unsigned*>(a.box[a.me]) + t * kPeerFlagStride;
    const unsigned long long t0 = wall_clock64();
    // signed distance: sequence numbers may wrap
    while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - a.seq) < 0) {
      if (wall_clock64() - t0 > a.budget) {   // 5 s: a rank is gone; end the launch, the host raises
        __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope; the payload loads below are system-scope loads of uncached memory as well
  if (n_red > 0) {
    const double* mine = reinterpret_cast<const double*>(a.box[a.me] + kPeerRedOff) + (size_t)par * a.size * a.red_cap;
    for (int i = t; i < n_red; i += nt) {
      double acc = 0.0;
      for (int q = 0; q < a.size; ++q) acc += ld_sys(mine + (size_t)q * a.red_cap + i);
      red_dst[i] = acc;
    }
  }
  if (per > 0)
    for (int j = 0; j < a.n_nb; ++j) {
      const int len = a.recv_ptr[j + 1] - a.recv_ptr[j], n = len * per;
      const double* src = reinterpret_cast<const double*>(a.box[a.me] + a.halo_off) + ((size_t)2 * a.recv_ptr[j] + (size_t)par * len) * a.wmax;
      double* dst = recvbuf + (size_t)a.recv_ptr[j] * per;
      for (int i = t; i < n; i += nt) dst[i] = ld_sys(src + i);
    }
}

// The per-rank sums of a BiCGStab half-iteration, the ghost rows that follow it AND their exchange over the peer mailboxes in ONE
// launch (instead of k_dist_reduce_pack + k_peer_exchange + the unpacking that rode in the next coarse launch):
//   workgroups [0, nout)  reduce one output value each (dist_reduce_block) and publish it write-through;
//   the others            store the rows of (up to three) vectors at the rank's boundary nodes STRAIGHT into the neighbours'
//                         mailboxes — no send buffer;
//   every workgroup       fences (system scope), then arrives at a counter; the one that arrives LAST does the rest of
//                         k_peer_exchange: contributions to every mailbox, flags, wait, sum in rank order into `out`, and the
//                         received rows into the ghost rows of the same vectors.
__global__ __launch_bounds__(256) void k_dist_reduce_exchange(const Ctx c, int phase, int par, double* __restrict__ out, int nout, const VecListW vecs, int nvec,
                                                               int width, const int32_t* __restrict__ send_nodes, int nsn,
                                                               const int32_t* __restrict__ recv_nodes, const PeerArgs a, unsigned* __restrict__ counter) {
  __shared__ int last_flag;
  const int t = threadIdx.x, per = nvec * width, slot = a.seq & 1;
  if ((int)blockIdx.x < nout) {
    dist_reduce_block<true>(c, phase, par, out, blockIdx.x);   // write-through: the last workgroup (any XCD) reads it back with agent-scope loads
  } else {
    const int i = ((int)blockIdx.x - nout) * 256 + t;
    if (i < nsn * per) {
      const int f = i % width, v = (i / width) % nvec, k = i / per;
      int j = 0;
      while (j + 1 < a.n_nb && k >= a.send_ptr[j + 1]) ++j;
      const int len = a.send_ptr[j + 1] - a.send_ptr[j];
      double* dst = reinterpret_cast<double*>(a.box[a.nb_rank[j]] + a.halo_off) + ((size_t)2 * a.peer_recv_ptr[j] + (size_t)slot * len) * a.wmax;
      st_sys(dst + (size_t)(k - a.send_ptr[j]) * per + v * width + f, vecs.p[v][(size_t)send_nodes[k] * width + f]);
    }
  }
  // No fence here: every store above is a write-through store (agent / system scope) and each thread drains its own; a
  // system-scope fence in each of the ~250 workgroups writes back and invalidates the XCD's L2 every time (measured: 80 us per
  // BiCGStab iteration instead of 40 at world size 1).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // (relaxed: the arrivals are ordered behind the drained stores by the wait and the barrier above; an acq_rel add is a release
  // fence — an L2 write-back — in every workgroup)
  if (t == 0) last_flag = (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!last_flag) return;
  if (t == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
  const int nt = 256;
  for (int q = 0; q < a.size; ++q) {
    double* dst = reinterpret_cast<double*>(a.box[q] + kPeerRedOff) + ((size_t)slot * a.size + a.me) * a.red_cap;
    for (int i = t; i < nout; i += nt) st_sys(dst + i, __hip_atomic_load(out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // Every byte of the messages went out as a write-through system-scope store and has been waited for by its thread (the
  // other workgroups': before their arrival at the counter).  The flags follow as RELEASE stores at system scope and the
  // wait ends in an ACQUIRE fence — in this one workgroup only, i.e. one release and one acquire per launch (the same pair in
  // every workgroup cost 20 us per iteration); the receivers read flags and payload with system-scope loads of uncached memory.
  if (t < a.size)
    __hip_atomic_store(reinterpret_cast<unsigned*>(a.box[t]) + a.me * kPeerFlagStride, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (t < a.size) {
    const unsigned* fl = reinterpret_cast<const unsigned*>(a.box[a.me]) + t * kPeerFlagStride;
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - a.seq) < 0) {
      if (wall_clock64() - t0 > a.budget) { __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope
  {
    const double* mine = reinterpret_cast<const double*>(a.box[a.me] + kPeerRedOff) + (size_t)slot * a.size * a.red_cap;
    for (int i = t; i < nout; i += nt) {
      double acc = 0.0;
      for (int q = 0; q < a.size; ++q) acc += ld_sys(mine + (size_t)q * a.red_cap + i);
      __hip_atomic_store(out + i, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through too: `out` is never dirty in an L2
    }
  }
  for (int j = 0; j < a.n_nb; ++j) {
    const int len = a.recv_ptr[j + 1] - a.recv_ptr[j], n = len * per;
    const double* src = reinterpret_cast<const double*>(a.box[a.me] + a.halo_off) + ((size_t)2 * a.recv_ptr[j] + (size_t)slot * len) * a.wmax;
    for (int i = t; i < n; i += nt) {
      const int f = i % width, v = (i / width) % nvec, k = a.recv_ptr[j] + i / per;
      vecs.p[v][(size_t)recv_nodes[k] * width + f] = ld_sys(src + i);
    }
  }
}

// ---- the exchange as the PROLOGUE of the next half-iteration's launch (peer transport, launches resident at once) --------------
// k_dist_reduce_exchange between two fused half-iteration launches costs its 7 us (measured phase by phase, tools/xch_phases.py:
// 1.3 us to sum the partials, 1.9 to drain the stores into the uncached mailboxes, 1.3 for the arrival counter, 0.5 + 0.8 for the
// flags, 1.0 for the acquire fence, 1.1 to read the contributions back) and sits between the launches.  Here the exchange rides IN
// FRONT of the coarse workgroups of the launch that needs it, and what travels is FLAGGED WORDS (ll_store / ll_wait in
// gmpnp_kernels.h): no flag behind the data, hence no drain, no counter, no elected workgroup, no fence —
//     [ nx exchange workgroups | nagg coarse workgroups | tiles ]
//   exchange workgroups   one WAVE per output value sums the previous launch's partials (plain loads: a kernel boundary lies in
//                         between) and stores the sum as flagged words into EVERY rank's mailbox; the other workgroups store the
//                         boundary rows of the previous launch's vectors as flagged words into the neighbours' mailboxes.  That is
//                         all: they wait for nobody.
//   coarse workgroups     issue their loads, then poll exactly the words they need in this rank's mailbox — the contributions of
//                         every rank to the few sums of their aggregate and to the scalars, added in rank order — and go on as ever;
//   tiles                 as on one GPU; a tile with ghost columns reads those entries out of the mailbox behind the hand-over.
// The sums and rows of an exchange stay in their slot (sequence number & 3) until the exchange after the next overwrites it: a
// launch also reads what the exchange BEFORE its own delivered (half A: P^T v, r, p and the ghost rows of p, v), so nothing is ever
// copied into red_a / red_b or into the vectors' ghost rows.  A rank can be one exchange ahead of a neighbour (it sends before it
// waits) while that neighbour still reads the exchange before: three live slots, four provided.  Nothing waits for a higher block
// index, so the launch drains whatever the residency.  k = 0 and everything outside the BiCGStab loop use the flag-based
// launches above (their own sequence numbers and mailbox areas).
struct XchArgs {
  unsigned char* box[kPeerMax];           // every rank's mailbox as mapped in this process
  int me, size;
  unsigned seq;                           // number of this flagged-word exchange: the same on every rank, consecutive within a solve
  size_t ll_red_off, ll_halo_off;         // byte offsets of the two flagged-word areas (the same on every rank)
  int red_cap;
  int n_nb, nb_rank[kPeerNbMax], send_ptr[kPeerNbMax + 1], peer_recv_ptr[kPeerNbMax];
  int phase, par, nout, nvec, nsn, nx;    // nx = exchange workgroups = ceil(nout / 8) + ceil(nsn nvec NF / 512)
  VecList vecs;
  const int32_t* send_nodes;
};
inline int xch_workgroups(int nout, int nsn, int nvec, int nf) { return (nout + 7) / 8 + (nsn * nvec * nf + kKrylovThreads - 1) / kKrylovThreads; }

template <int NF>
__device__ __forceinline__ void xch_body(const Ctx& c, const XchArgs& x, const int wg, const int k) {
  static_assert(3 * NF <= kLLRow, "three vectors of a ghost node per slot");
  constexpr int nt = kKrylovThreads;
  const int t = threadIdx.x, per = x.nvec * NF, slot = x.seq & (kLLSlots - 1);
  const int nred = (x.nout + 7) / 8;
  if (c.scal->done) return;   // a launch behind the end of the solve: nobody reads what it would send (the verdict is the same on every rank)
  if (wg == 0 && x.phase == 2) GMPNP_XSTAMP(k, 0);
  if (wg < nred) {
    const int o = wg * 8 + (t >> 6), lane = t & 63;
    if (o < x.nout) {   // wave-uniform
      int count, stride;
      const double* p = dist_reduce_source(c, x.phase, x.par, o, count, stride);
      // twelve requests per lane in flight at once: the launches this form is used for are resident at once, i.e. at most
      // 768 tiles (and as many slots) — one memory round trip for the whole sum, the loop only for anything larger
      double v = 0.0;
      for (int i0 = lane; i0 < count; i0 += 12 * 64) {
        double w12[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) w12[u] = p[(size_t)min(i0 + u * 64, count - 1) * stride];
#pragma unroll
        for (int u = 0; u < 12; ++u) v += (i0 + u * 64 < count) ? w12[u] : 0.0;
      }
      v = wave_sum(v);
      if (wg == 0 && x.phase == 2) GMPNP_XSTAMP(k, 1);
      if (lane < x.size)
        ll_store(reinterpret_cast<unsigned long long*>(x.box[lane] + x.ll_red_off) + (((size_t)slot * x.size + x.me) * x.red_cap + o) * 2, v, x.seq);
    }
  } else {
    const int i = (wg - nred) * nt + t;
    if (i < x.nsn * per) {
      const int f = i % NF, v = (i / NF) % x.nvec, ks = i / per;
      int j = 0;
      while (j + 1 < x.n_nb && ks >= x.send_ptr[j + 1]) ++j;
      const int kr = x.peer_recv_ptr[j] + (ks - x.send_ptr[j]);   // the node's index in the NEIGHBOUR's receive list
      ll_store(reinterpret_cast<unsigned long long*>(x.box[x.nb_rank[j]] + x.ll_halo_off) + (((size_t)kr * kLLSlots + slot) * kLLRow + v * NF + f) * 2,
               x.vecs.p[v][(size_t)x.send_nodes[ks] * NF + f], x.seq);
    }
  }
  if (wg == 0 && x.phase == 2) GMPNP_XSTAMP(k, 2);
}

template <int NF>
__global__ __launch_bounds__(kKrylovThreads, 6) void k_half_a_x(const Ctx c, const int k, const unsigned target, const XchArgs x) {
  const int b = (int)blockIdx.x - x.nx;
  if (b < 0) xch_body<NF>(c, x, blockIdx.x, k);
  else if (b < c.nagg) coarse_a_body<NF, true, true>(c, k, b, target);
  else bicg_a_body<NF, true, false, true>(c, k, c.tile0 + xcd_tile(b - c.nagg, (int)gridDim.x - x.nx - c.nagg), target);
}
template <int NF>
__global__ __launch_bounds__(kKrylovThreads, 6) void k_half_b_x(const Ctx c, const int k, const unsigned target, const XchArgs x) {
  const int b = (int)blockIdx.x - x.nx;
  if (b < 0) xch_body<NF>(c, x, blockIdx.x, k);
  else if (b < c.nagg) coarse_b_body<NF, true, true>(c, k, b, target);
  else bicg_b_body<NF, true, false, true>(c, k, c.tile0 + xcd_tile(b - c.nagg, (int)gridDim.x - x.nx - c.nagg), target);
}

// Self-check of the flagged-word areas over the transport itself (gmpnp_group_selftest): every rank stores (rank + 1)(i + 1), i < 5, into
// every rank's sums area and sender * 1e6 + k into the first value of node k of every neighbour's ghost-row area, then polls what the
// others sent and compares.  err_out[0] = largest deviation seen by this rank (a word that never arrives: status bit 8 as in a solve,
// and the deviation of whatever stands there).  One workgroup.
__global__ __launch_bounds__(kKrylovThreads) void k_xch_selftest(const Ctx c, const XchArgs x, const PeerArgs a, double* __restrict__ err_out) {
  __shared__ double lds[kKrylovThreads / 64];
  const int t = threadIdx.x, slot = x.seq & (kLLSlots - 1);
  if (t < 5)
    for (int q = 0; q < x.size; ++q)
      ll_store(reinterpret_cast<unsigned long long*>(x.box[q] + x.ll_red_off) + (((size_t)slot * x.size + x.me) * x.red_cap + t) * 2, (double)(x.me + 1) * (t + 1), x.seq);
  for (int j = 0; j < x.n_nb; ++j)
    for (int ks = x.send_ptr[j] + t; ks < x.send_ptr[j + 1]; ks += kKrylovThreads) {
      const int kr = x.peer_recv_ptr[j] + (ks - x.send_ptr[j]);
      ll_store(reinterpret_cast<unsigned long long*>(x.box[x.nb_rank[j]] + x.ll_halo_off) + (((size_t)kr * kLLSlots + slot) * kLLRow) * 2, 1e6 * x.me + (ks - x.send_ptr[j]), x.seq);
    }
  double err = 0.0;
  if (t < 5) {
    double acc = 0.0;
    for (int q = 0; q < x.size; ++q) {
      const unsigned long long* const src[1] = {ll_red_word(c, x.seq, q, t)};
      const uint32_t sq[1] = {x.seq};
      double v[1];
      ll_wait_n<1>(c, src, sq, v);
      acc += v[0];
    }
    err = fabs(acc - 0.5 * x.size * (x.size + 1) * (t + 1));
  }
  for (int j = 0; j < a.n_nb; ++j)
    for (int k = a.recv_ptr[j] + t; k < a.recv_ptr[j + 1]; k += kKrylovThreads) {
      const unsigned long long* const src[1] = {ll_ghost_word(c, x.seq, k, 0, 0, 9)};
      const uint32_t sq[1] = {x.seq};
      double v[1];
      ll_wait_n<1>(c, src, sq, v);
      err = fmax(err, fabs(v[0] - (1e6 * a.nb_rank[j] + (k - a.recv_ptr[j]))));
    }
  // largest deviation of the workgroup (max over the lanes by shuffles, over the waves through LDS)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) err = fmax(err, __shfl_xor(err, o, 64));
  if ((t & 63) == 0) lds[t >> 6] = err;
  __syncthreads();
  if (t == 0) { double m = 0.0; for (int w = 0; w < kKrylovThreads / 64; ++w) m = fmax(m, lds[w]); err_out[0] = m; }
}

// In-process rehearsal transport: sum over the handles of one process, written back to all of them (fixed order)
struct PtrList { double* p[8]; };
__global__ __launch_bounds__(256) void k_local_allreduce(const PtrList bufs, int nb, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int b = 0; b < nb; ++b) s += bufs.p[b][i];
  for (int b = 0; b < nb; ++b) bufs.p[b][i] = s;
}

}  // namespace gmpnp
