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
//   k_band_step     pivot k: A_pq -= A_pk (D_k^-1 A_kq) for the (<= b)^2 window behind it; one extra workgroup: D_{k+1}^-1.  Row k and
//                   column k are only READ (they are the factors: L_pk = A_pk, U_kq = D_k^-1 A_kq with unit diagonal),
//                   everything written lies strictly behind them: no hazard inside a launch, one launch per pivot.
//   k_band_panel / k_band_tri   forward and backward substitution in panels of 16 block rows (see below)
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

// grid (ceil(w / kBandColChunk), 1 + ceil(w / G)), w = min(b, n-1-k); k = -1: grid (1, 1).  The FIRST grid row is one extra
// workgroup (blockIdx.x = 0; the others of that row leave at once) that owns block (k+1, k+1): it applies pivot k to that one
// block, stores it and inverts it — the NEXT launch's pivot — while the window is being updated by everybody else.  Done at
// the end of the launch by the workgroup that holds the block, the dependent inverse (4 us of 16-lane shuffles) was a tail
// every launch waited for.  k = -1: only D_0^-1.
template <int NF>
__global__ __launch_bounds__(kBandThreads) void k_band_step(const BandLU lu, const int k, int32_t* status) {
  constexpr int BB = NF * NF, G = kBandThreads / BB, QC = kBandColChunk;
  __shared__ double sU[QC * BB];
  __shared__ double sD[BB];
  const int t = threadIdx.x, g = t / BB, e = t - g * BB, i = e / NF, j = e - i * NF;
  if (blockIdx.y == 0) {   // the pivot workgroup: FIRST grid row, so that it is dispatched with the first workgroups whatever the window size
    if (blockIdx.x != 0) return;
    double entry = 0.0;
    if (k >= 0) {
      const int tc = min(t, BB - 1), ic = tc / NF, jc = tc - ic * NF;
      double a9[NF], l9[NF];
      const double* Ak = band_at<NF>(lu, k, k + 1);
      const double* Lp = band_at<NF>(lu, k + 1, k) + ic * NF;
#pragma unroll
      for (int m = 0; m < NF; ++m) { a9[m] = Ak[m * NF + jc]; l9[m] = Lp[m]; }
      double* Cd = band_at<NF>(lu, k + 1, k + 1) + tc;
      double acc = *Cd;
      if (t < BB) sD[t] = lu.dinv[(size_t)k * BB + t];
      __syncthreads();
      double u = 0.0;   // same operation order as the window update below
#pragma unroll
      for (int ll = 0; ll < NF; ++ll) u += sD[ic * NF + ll] * a9[ll];
      if (t < BB) sU[t] = u;
      __syncthreads();
#pragma unroll
      for (int m = 0; m < NF; ++m) acc -= l9[m] * sU[m * NF + jc];
      if (t < BB) *Cd = acc;
      entry = acc;
    } else if (t < BB) {
      entry = band_at<NF>(lu, 0, 0)[t];
    }
    if (k + 1 >= lu.n) return;
    // D_{k+1}^-1 (sD is free: its last readers passed the barrier above)
    if (t < BB) sD[t] = entry;
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
    return;
  }
  const int w = min(lu.b, lu.n - 1 - k);
  const int q0 = k + 1 + blockIdx.x * QC, p = k + 1 + ((int)blockIdx.y - 1) * G + g;
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
    const bool pivot_block = (p == k + 1 && q0 == k + 1);   // (k+1, k+1) belongs to the pivot workgroup
#pragma unroll
    for (int qq = 0; qq < QC; ++qq) {
      double acc = cv[qq];
#pragma unroll
      for (int m = 0; m < NF; ++m) acc -= l[m] * sU[min(qq, nq - 1) * BB + m * NF + j];
      if (qq < nq && !(pivot_block && qq == 0)) C[(size_t)qq * BB] = acc;
    }
  }
}

// ---- substitution: panels of kBandPanel block rows, two launches per panel -----------------------------------------
// x = U^-1 L^-1 rhs with y_k = D_k^-1 (rhs_k - sum_{q<k} A_kq y_q) and x_k = y_k - D_k^-1 sum_{q>k} A_kq x_q.  The chain
// over the n block rows is what one workgroup used to walk alone (118 KB of band row per step through one CU: 27 ms on
// L_50_R_5).  Now the band row of a block row is split at the panel boundary:
//   k_band_panel   everything OUTSIDE the panel (blocks solved by earlier panels): one workgroup per (block row, chunk of
//                  the band row), NF waves = the NF rows of the block, partial sums to `part` (summed in fixed order by the
//                  next kernel: bitwise repeatable, no atomics);
//   k_band_tri     the kBandPanel x kBandPanel block triangle INSIDE the panel: one workgroup, the whole triangle
//                  requested into registers before the chain starts, D_k^-1 and the panel's vectors in LDS; a step is
//                  products -> wave sums -> barrier -> D_k^-1 (NF threads) -> barrier.
// y holds rhs in elimination order on entry of the forward sweep, y after it and x (elimination order) after the backward
// sweep, which also scatters x to internal node order.
constexpr int kBandPanel = 16;
constexpr int kBandChunks = 4;      // chunks of the outside part of a band row (one workgroup each)
constexpr int kBandDotBatch = 16;   // band-row entries per lane requested together (clamped indices, masks afterwards)

template <int NF>
__global__ __launch_bounds__(256) void k_band_gather(const BandLU lu, const double* __restrict__ rhs, double* __restrict__ y) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= lu.n * NF) return;
  const int k = idx / NF;
  y[idx] = rhs[(size_t)lu.lu_node[k] * NF + (idx - k * NF)];
}

// grid (K1 - K0, kBandChunks).  FWD: blocks [max(0, k-b), K0) of block row k; backward: blocks [K1, min(k+b, n-1)].
template <int NF, bool FWD>
__global__ __launch_bounds__(NF * kWave) void k_band_panel(const BandLU lu, const double* __restrict__ y, double* __restrict__ part,
                                                            const int K0, const int K1) {
  constexpr int U = kBandDotBatch, BB = NF * NF;
  const int t = threadIdx.x, wv = t / kWave, lane = t - wv * kWave;
  const int kk = blockIdx.x, ch = blockIdx.y, k = K0 + kk;
  const int q_lo = FWD ? max(0, k - lu.b) : K1, q_hi = FWD ? K0 : min(k + lu.b, lu.n - 1) + 1;
  const int nblk = max(q_hi - q_lo, 0), per = (nblk + kBandChunks - 1) / kBandChunks;
  const int c_lo = q_lo + ch * per, c_hi = min(c_lo + per, q_hi);
  const int count = max(c_hi - c_lo, 0) * NF, last = max(count - 1, 0);
  double acc = 0.0;
  if (count > 0) {   // uniform per workgroup
    const double* row = band_at<NF>(lu, k, c_lo) + wv * NF;   // row wv of the blocks: NF entries every BB
    const double* yv = y + (size_t)c_lo * NF;                 // entry (qo, j) of the row meets y[(c_lo + qo) * NF + j]
    for (int base = 0; base < count; base += U * kWave) {
      double w[U], z[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = min(base + u * kWave + lane, last), qo = idx / NF, jj = idx - qo * NF;
        w[u] = row[(size_t)qo * BB + jj]; z[u] = yv[idx];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += (base + u * kWave + lane < count) ? w[u] * z[u] : 0.0;
    }
  }
  acc = wave_sum(acc);
  if (lane == kWave - 1) part[((size_t)kk * kBandChunks + ch) * NF + wv] = acc;
}

// one workgroup; np = K1 - K0 <= kBandPanel block rows.  Step s solves block row K0 + s (forward) / K1 - 1 - s (backward):
// s blocks of the panel stand in its band row either way.
template <int NF, bool FWD>
__global__ __launch_bounds__(NF * kWave) void k_band_tri(const BandLU lu, double* __restrict__ y, const double* __restrict__ part,
                                                          double* __restrict__ x, const int K0, const int K1) {
  constexpr int P = kBandPanel, BB = NF * NF, UM = ((P - 1) * NF + kWave - 1) / kWave;
  __shared__ double yp[P * NF];   // the panel's solution blocks
  __shared__ double dl[P * BB];   // D_k^-1
  __shared__ double rh[P * NF];   // forward: rhs_k - outside sum; backward: outside sum
  __shared__ double yk[P * NF];   // backward: y_k
  __shared__ double st[NF];
  const int t = threadIdx.x, wv = t / kWave, lane = t - wv * kWave;
  const int np = K1 - K0;
  // every global operand first: the triangle (registers), then D^-1, the outside sums and the panel's y
  double v[P][UM];
#pragma unroll
  for (int s = 1; s < P; ++s) {
    const int sc = min(s, np - 1), k = FWD ? K0 + sc : K1 - 1 - sc;
    const double* row = (FWD ? band_at<NF>(lu, k, K0) : band_at<NF>(lu, k, k + 1)) + wv * NF;
    const int last = max(sc * NF - 1, 0);
#pragma unroll
    for (int u = 0; u < UM; ++u)
      if (u * kWave < s * NF) {   // compile time
        const int idx = min(u * kWave + lane, last), qo = idx / NF, jj = idx - qo * NF;
        v[s][u] = (np > 1) ? row[(size_t)qo * BB + jj] : 0.0;
      }
  }
  for (int idx = t; idx < np * BB; idx += NF * kWave) dl[idx] = lu.dinv[(size_t)K0 * BB + idx];
  if (t < np * NF) {
    const int kk = t / NF;
    double sacc = 0.0;
#pragma unroll
    for (int c = 0; c < kBandChunks; ++c) sacc += part[((size_t)kk * kBandChunks + c) * NF + (t - kk * NF)];
    const double yv = y[(size_t)K0 * NF + t];
    rh[t] = FWD ? yv - sacc : sacc;
    if (!FWD) yk[t] = yv;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < P; ++s) {
    if (s < np) {   // uniform
      const int kk = FWD ? s : np - 1 - s;
      const double* ys = yp + (FWD ? 0 : (kk + 1) * NF);   // entry (qo, j) of the in-panel row meets ys[qo * NF + j]
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < UM; ++u)
        if (u * kWave < s * NF) {   // compile time
          const int raw = u * kWave + lane;
          acc += (raw < s * NF) ? v[s][u] * ys[min(raw, s * NF - 1)] : 0.0;
        }
      if (s > 0) acc = wave_sum(acc);
      if (lane == kWave - 1) st[wv] = FWD ? rh[kk * NF + wv] - acc : rh[kk * NF + wv] + acc;
      __syncthreads();
      if (t < NF) {
        double r = 0.0;
#pragma unroll
        for (int m = 0; m < NF; ++m) r += dl[kk * BB + t * NF + m] * st[m];
        if (!FWD) r = yk[kk * NF + t] - r;
        yp[kk * NF + t] = r;
      }
      __syncthreads();
    }
  }
  // results leave in one piece (a global store inside the chain would be drained at every barrier)
  if (t < np * NF) {
    const int kk = t / NF;
    const double r = yp[t];
    y[(size_t)K0 * NF + t] = r;
    if (!FWD) x[(size_t)lu.lu_node[K0 + kk] * NF + (t - kk * NF)] = r;
  }
}

}  // namespace gmpnp
