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
__device__ __forceinline__ void dist_reduce_block(const Ctx& c, int phase, int par, double* __restrict__ out, int o) {
  __shared__ double lds[4];
  const int n = c.ncoarse, t = threadIdx.x;
  const int nscal = phase == 1 ? 2 : (phase == 2 ? 4 : 0);
  const double* p; int count, stride;
  if (o < nscal) {
    p = phase == 1 ? (o == 0 ? c.part_a : c.part_rr) : c.part_b + (size_t)o * c.ntiles;
    count = c.ntiles; stride = 1;
  } else {
    const int q = o - nscal, w = q / n, d = q - w * n;
    p = (phase == 0 ? c.cpart_v[1] : phase == 3 ? c.cpart_v[0] : phase == 2 ? c.cpart_t
         : (w == 0 ? c.cpart_v[par] : w == 1 ? c.cpart_r[par] : c.cpart_p[par])) + d;
    count = c.tile_slots; stride = n;
  }
  double v[1] = {0.0};
  for (int i0 = t; i0 < count; i0 += 8 * 256) {   // eight independent requests per thread and trip
    double w8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w8[u] = p[(size_t)min(i0 + u * 256, count - 1) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[0] += (i0 + u * 256 < count) ? w8[u] : 0.0;
  }
  block_sum<1>(v, lds);
  if (t == 0) out[o] = v[0];
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
