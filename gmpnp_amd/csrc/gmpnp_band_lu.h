// Block-banded LU of the node-block Jacobian on gfx950: the DIRECT linear solver of the 3D path.
//
// The reference solves every Newton system with MUMPS (3D/MPNP_CO2ER_pore.py:792, PETSc KSP preonly + LU).  The fast
// path of this library is the two-level BiCGStab; this file is what takes over when that Krylov solve does not converge
// (the stiff systems of the thinnest pores), and what `GMPNP_LINEAR_BAND_LU` selects explicitly.
//
// Elimination order = the caller's slab order (vertices sorted along the pore axis): in that order the pattern is a
// band of b node blocks (182 on L_50_R_5), which the elimination fills completely, so the factor is stored as a dense
// block band (BandLU in gmpnp_internal.h).  No MFMA: the update is a rank-NF (9) product per pivot, HBM-bound.
//
//   k_band_scatter  SELL values -> band storage
//   k_band_step     pivot k: A_pq -= A_pk (D_k^-1 A_kq) for the (<= b)^2 window behind it, then D_{k+1}^-1.  Row k and
//                   column k are only READ (they are the factors: L_pk = A_pk, U_kq = D_k^-1 A_kq with unit diagonal),
//                   everything written lies strictly behind them: no hazard inside a launch, one launch per pivot.
//   k_band_solve    forward and backward substitution, one workgroup walking the band rows (row dot products)
//
// Pivoting is partial INSIDE the NF x NF pivot block only; the caller checks the answer with the true residual
// b - J x and refines (gmpnp_api.hip: band_solve).
#pragma once
#include "gmpnp_kernels.h"

namespace gmpnp {

constexpr int kBandThreads = 512;
constexpr int kBandColChunk = 16;   // pivot-row blocks one workgroup multiplies by D_k^-1 and keeps in LDS

template <int NF>
__device__ __forceinline__ double* band_at(const BandLU& lu, int p, int q) {
  return lu.band + ((size_t)p * (2 * lu.b + 1) + (size_t)(q - p + lu.b)) * (NF * NF);
}

template <int NF>
__global__ __launch_bounds__(64) void k_band_scatter(const Ctx c, const BandLU lu) {
  const int s = blockIdx.x, lane = threadIdx.x;
  if (lane >= c.slice_nn[s] * NF) return;
  const int il = lane / NF, i = lane - il * NF;
  const int I = c.slice_node0[s] + il, p = lu.lu_pos[I];
  const int cb = c.slice_colbase[s], mx = c.slice_colbase[s + 1] - cb;
  const double* v = c.vals + c.slice_off[s] + lane;
  for (int kp = 0; kp < mx; ++kp) {
    const size_t rec = (size_t)(cb + kp) * kSlicePad + il;
    if (c.sell_blk[rec] < 0) continue;
    const int q = lu.lu_pos[c.sell_cols[rec] & 0xffffff];
    double* dst = band_at<NF>(lu, p, q) + i * NF;
#pragma unroll
    for (int j = 0; j < NF; ++j) dst[j] = v[(size_t)(kp * NF + j) * kWave];
  }
}

// Gauss-Jordan inverse of one NF x NF block by a 16-lane group: lane r (< NF) holds row r of [A | I].
// Partial pivoting; returns true when a pivot column was all zero / NaN.
template <int NF>
__device__ inline bool group16_inverse(double (&row)[2 * NF], int r) {
  bool bad = false;
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    double v = (r < NF && r >= k) ? fabs(row[k]) : -1.0; int idx = r;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const double ov = __shfl_xor(v, o, 16); const int oi = __shfl_xor(idx, o, 16);
      if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    bad |= !(v > 0.0);
    const double ip = 1.0 / __shfl(row[k], idx, 16);
    const double oldk_k = __shfl(row[k], k, 16);
    const double f = ((r == idx) ? oldk_k : row[k]) * ip;
#pragma unroll
    for (int j = k + 1; j < 2 * NF; ++j) {
      const double from_p = __shfl(row[j], idx, 16), from_k = __shfl(row[j], k, 16);
      const double mine = (r == idx) ? from_k : row[j];
      row[j] = (r == k) ? from_p * ip : mine - f * from_p;
    }
    row[k] = (r == k) ? 1.0 : 0.0;
  }
  return bad;
}

// grid (ceil(w / kBandColChunk), ceil(w / G)), w = min(b, n-1-k) ; k = -1: grid (1,1), only D_0^-1.
template <int NF>
__global__ __launch_bounds__(kBandThreads) void k_band_step(const BandLU lu, const int k, int32_t* status) {
  constexpr int BB = NF * NF, G = kBandThreads / BB, QC = kBandColChunk;
  __shared__ double sU[QC * BB];
  __shared__ double sD[BB];
  const int t = threadIdx.x, g = t / BB, e = t - g * BB, i = e / NF, j = e - i * NF;
  const bool first_wg = (blockIdx.x == 0 && blockIdx.y == 0);
  double next_piv = 0.0;   // entry e of block (k+1, k+1) after this pivot's update (group 0 of the first workgroup)
  if (k >= 0) {
    const int w = min(lu.b, lu.n - 1 - k);
    const int q0 = k + 1 + blockIdx.x * QC, p = k + 1 + blockIdx.y * G + g;
    const int nq = min(QC, k + w + 1 - q0);
    // every global operand is requested before the first barrier (clamped indices, masks afterwards): one memory
    // latency per launch instead of three dependent ones; the launches of a factorisation run back to back
    constexpr int UP = (QC * BB + kBandThreads - 1) / kBandThreads;   // U' entries per thread
    const bool act = g < G && p <= k + w;
    const int pc = min(p, k + w);
    const double dk = lu.dinv[(size_t)k * BB + min(t, BB - 1)];
    const double* Ak = band_at<NF>(lu, k, q0);
    double a[UP][NF];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const int idx = min(t + u * kBandThreads, nq * BB - 1), qq = idx / BB, ee = idx - qq * BB, jj = ee % NF;
#pragma unroll
      for (int l = 0; l < NF; ++l) a[u][l] = Ak[(size_t)qq * BB + l * NF + jj];
    }
    const double* Lp = band_at<NF>(lu, pc, k) + i * NF;
    double l[NF];
#pragma unroll
    for (int m = 0; m < NF; ++m) l[m] = Lp[m];
    double* C = band_at<NF>(lu, pc, q0) + e;
    double cv[QC];
#pragma unroll
    for (int qq = 0; qq < QC; ++qq) cv[qq] = C[(size_t)min(qq, nq - 1) * BB];
    if (t < BB) sD[t] = dk;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const int idx = t + u * kBandThreads;
      const int ic = min(idx, nq * BB - 1), qq = ic / BB, ee = ic - qq * BB, m = ee / NF;
      double acc = 0.0;
#pragma unroll
      for (int ll = 0; ll < NF; ++ll) acc += sD[m * NF + ll] * a[u][ll];
      if (idx < nq * BB) sU[idx] = acc;
    }
    __syncthreads();
    if (act) {
#pragma unroll
      for (int qq = 0; qq < QC; ++qq) {
        double acc = cv[qq];
#pragma unroll
        for (int m = 0; m < NF; ++m) acc -= l[m] * sU[min(qq, nq - 1) * BB + m * NF + j];
        if (qq < nq) C[(size_t)qq * BB] = acc;
        if (qq == 0) next_piv = acc;
      }
    }
  } else if (t < BB) {
    next_piv = band_at<NF>(lu, 0, 0)[t];
  }
  if (first_wg && k + 1 < lu.n) {   // uniform per workgroup
    __syncthreads();
    if (t < BB) sD[t] = next_piv;
    __syncthreads();
    if (t < kWave) {   // wave 0: its four 16-lane groups all do the same work, group 0 writes
      const int r = t & 15;
      double row[2 * NF];
#pragma unroll
      for (int jj = 0; jj < NF; ++jj) { row[jj] = (r < NF) ? sD[(r < NF ? r : 0) * NF + jj] : 0.0; row[NF + jj] = (jj == r) ? 1.0 : 0.0; }
      const bool bad = group16_inverse<NF>(row, r);
      if (t < NF) {
        double* o = lu.dinv + ((size_t)(k + 1) * NF + t) * NF;
#pragma unroll
        for (int jj = 0; jj < NF; ++jj) o[jj] = row[NF + jj];
      }
      if (t == 0 && bad) atomicOr(status, 2);
    }
  }
}

// One lane's share of a band row (entries (qo, j) of the qo-th block behind `row`, row stride already applied), in
// batches of kBandDotBatch entries per lane: all loads of a batch go out on clamped indices before the first use, the
// masks come afterwards (a plain loop waits for every load before it issues the next: one memory latency per 64
// entries; 54 -> 27 ms per substitution on the pore meshes).  Requesting the first batch of the NEXT row one step
// ahead was tried and lost: two batches of doubles in flight per lane spill at 9 waves per workgroup.
constexpr int kBandDotBatch = 16;
template <int NF>
__device__ __forceinline__ void band_row_load(const double* __restrict__ row, int count, int base, int lane, double (&v)[kBandDotBatch]) {
  constexpr int BB = NF * NF;
  const int last = max(count - 1, 0);
#pragma unroll
  for (int u = 0; u < kBandDotBatch; ++u) {
    const int idx = min(base + u * kWave + lane, last), qo = idx / NF, jj = idx - qo * NF;
    v[u] = row[(size_t)qo * BB + jj];
  }
}
template <int NF>
__device__ __forceinline__ double band_row_mac(const double (&v)[kBandDotBatch], const double* ring, int count, int base, int rb, int R, int lane) {
  const int last = max(count - 1, 0);
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < kBandDotBatch; ++u) {
    const int raw = base + u * kWave + lane, idx = min(raw, last), qo = idx / NF, jj = idx - qo * NF;
    int rr = rb + qo; rr -= (rr >= R) ? R : 0;
    acc += (raw < count) ? v[u] * ring[rr * NF + jj] : 0.0;
  }
  return acc;
}

// x = U^-1 L^-1 rhs.  ONE workgroup of NF waves; wave i owns row i of the current block row and reduces its band
// dot product, then NF threads apply D_k^-1.  The last b solution blocks live in an LDS ring ((b+1) * NF doubles,
// dynamic shared memory).  rhs and x are in internal node order, y is a work vector in elimination order.
// Every step is a chain band row -> dot -> D_k^-1 -> ring; whatever does not depend on the previous block (the band
// row, D_k^-1, the right-hand side) is requested ahead of the chain.
template <int NF>
__global__ __launch_bounds__(NF * kWave) void k_band_solve(const BandLU lu, const double* __restrict__ rhs, double* __restrict__ x,
                                                            double* __restrict__ y) {
  constexpr int U = kBandDotBatch;
  extern __shared__ double ring[];   // [(b+1)][NF]
  __shared__ double st[NF];
  const int t = threadIdx.x, wv = t / kWave, lane = t - wv * kWave;
  const int n = lu.n, b = lu.b, R = b + 1;
  for (int idx = t; idx < n * NF; idx += NF * kWave) {   // right-hand side in elimination order
    const int k = idx / NF;
    y[idx] = rhs[(size_t)lu.lu_node[k] * NF + (idx - k * NF)];
  }
  __syncthreads();
  const int tc = t < NF ? t : 0;
  // forward: y_k = D_k^-1 (rhs_k - sum_{q<k} A_kq y_q)
  for (int k = 0; k < n; ++k) {
    const int nb = min(b, k), qa = k - nb, count = nb * NF;
    const double* row = band_at<NF>(lu, k, qa) + wv * NF;
    const double* dptr = lu.dinv + ((size_t)k * NF + tc) * NF;
    double d[NF];
#pragma unroll
    for (int m = 0; m < NF; ++m) d[m] = dptr[m];
    const double bk = y[(size_t)k * NF + wv];
    const int rb = qa % R;
    double acc = 0.0;
    for (int base = 0; base < count; base += U * kWave) {
      double w[U];
      band_row_load<NF>(row, count, base, lane, w);
      acc += band_row_mac<NF>(w, ring, count, base, rb, R, lane);
    }
    acc = wave_sum(acc);
    if (lane == kWave - 1) st[wv] = bk - acc;
    __syncthreads();
    if (t < NF) {
      double r = 0.0;
#pragma unroll
      for (int m = 0; m < NF; ++m) r += d[m] * st[m];
      ring[(k % R) * NF + t] = r;
      y[(size_t)k * NF + t] = r;
    }
    __syncthreads();
  }
  // backward: x_k = y_k - D_k^-1 sum_{q>k} A_kq x_q
  for (int k = n - 1; k >= 0; --k) {
    const int nb = min(b, n - 1 - k), count = nb * NF;
    const double* row = band_at<NF>(lu, k, k + 1) + wv * NF;
    const double* dptr = lu.dinv + ((size_t)k * NF + tc) * NF;
    double d[NF];
#pragma unroll
    for (int m = 0; m < NF; ++m) d[m] = dptr[m];
    const double yk = y[(size_t)k * NF + tc];
    const int node = lu.lu_node[k];
    const int rb = (k + 1) % R;
    double acc = 0.0;
    for (int base = 0; base < count; base += U * kWave) {
      double w[U];
      band_row_load<NF>(row, count, base, lane, w);
      acc += band_row_mac<NF>(w, ring, count, base, rb, R, lane);
    }
    acc = wave_sum(acc);
    if (lane == kWave - 1) st[wv] = acc;
    __syncthreads();
    if (t < NF) {
      double r = yk;
#pragma unroll
      for (int m = 0; m < NF; ++m) r -= d[m] * st[m];
      ring[(k % R) * NF + t] = r;
      x[(size_t)node * NF + t] = r;
    }
    __syncthreads();
  }
}

}  // namespace gmpnp
