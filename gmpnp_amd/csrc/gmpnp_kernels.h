// HIP kernels of the GMPNP hot path for gfx950 (wave64, fp64 VALU; no MFMA: there is no dense contraction).
//
//   k_element      per-element P1 residual + the few scalars the exact Jacobian is built from     (a1-a4)
//   k_res_gather   race-free node gather of the residual, Dirichlet rows b = x - g, ||b||^2      (a5, a6)
//   k_jac_gather   race-free gather of the node-block Jacobian straight into SELL storage,
//                  identity Dirichlet rows                                                        (a5, a6)
//   k_spmv         SELL block SpMV with the coarse prolongation folded into the x gather          (a7)
//   k_vec1/k_vec2  fused BiCGStab vector updates + block-Jacobi apply + coarse restriction        (a7)
//   k_coarse       coarse GEMV with the LDS-inverted Galerkin operator                            (a7)
//
// Reference mathematics: 3D/MPNP_CO2ER_pore.py:505-769, 1D/MPNP_CO2ER_EDL.py:383-595.
#pragma once
#include "gmpnp_internal.h"

namespace gmpnp {

template <int DIM, int NF>
struct Lay {
  static constexpr int NS = NF - 1;
  static constexpr int NN = DIM + 1;
  static constexpr int S = kWave / NF;  // block rows per SELL slice
  // element-intermediate record (doubles)
  static constexpr int O_VOL = 0;
  static constexpr int O_GG = 1;                 // [NN][NN] grad phi_a . grad phi_b
  static constexpr int O_GP = O_GG + NN * NN;    // [NN] grad p . grad phi_a
  static constexpr int O_GG_A = O_GP + NN;       // [NN] G . grad phi_a
  static constexpr int O_UBAR = O_GG_A + NN;     // [NS]
  static constexpr int O_EPS = O_UBAR + NS;      // eps(ubar)
  static constexpr int O_IJ = O_EPS + 1;         // [NS] int u_i beta   (J rule)
  static constexpr int O_B = O_IJ + NS;          // [NN] int beta phi_b
  static constexpr int O_C = O_B + NN;           // [NS][NN] int u_i beta^2 phi_b
  static constexpr int O_U = O_C + NS * NN;       // [NN][NS] nodal species values (bilinear reaction derivative)
  static constexpr int EJ_STRIDE = O_U + NN * NS;
  static constexpr int EF_STRIDE = NN * NF;
  static constexpr double MDEN = 1.0 / ((DIM + 1) * (DIM + 2));  // M_ab = |K| (1+delta_ab) MDEN
  static constexpr double KAPPA = (DIM == 3) ? 1.0 / 120.0 : 1.0 / 24.0;  // d!/(d+3)!
};

// ---------------------------------------------------------------------------------------------
// deterministic workgroup reductions (256 threads = 4 waves)
// ---------------------------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int K>
__device__ inline void block_sum(double (&v)[K], double* lds /* [4*K] */) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < K; ++k) lds[w * K + k] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = (lds[k] + lds[K + k]) + (lds[2 * K + k] + lds[3 * K + k]);
}

// Sum K partial arrays (each n long, array k at part + k*stride) in a fixed order; every thread gets the totals.
template <int K>
__device__ inline void sum_partials(const double* __restrict__ part, int n, int stride, double (&out)[K], double* lds) {
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = 0.0;
  for (int i = threadIdx.x; i < n; i += kVecBlock)
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += part[(size_t)k * stride + i];
  block_sum<K>(v, lds);
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = v[k];
}

// ---------------------------------------------------------------------------------------------
// Element kernel: one lane per cell.
// ---------------------------------------------------------------------------------------------
template <int DIM, int NF, bool WANT_J>
__global__ __launch_bounds__(64) void k_element(const Ctx c) {
  using L = Lay<DIM, NF>;
  constexpr int NS = L::NS, NN = L::NN;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= c.nc) return;
  const gmpnp_model_t& m = *c.model;
  const gmpnp_quadrature_t& qd = *c.quad;

  int nd[NN];
  double X[NN][DIM], U[NN][NF];
#pragma unroll
  for (int a = 0; a < NN; ++a) {
    nd[a] = c.cells[e * NN + a];
#pragma unroll
    for (int d = 0; d < DIM; ++d) X[a][d] = c.coords[(size_t)nd[a] * DIM + d];
#pragma unroll
    for (int f = 0; f < NF; ++f) U[a][f] = c.u[(size_t)nd[a] * NF + f];
  }
  // geometry: |K| and the constant gradients of the P1 basis
  double g[NN][DIM], vol;
  if constexpr (DIM == 1) {
    const double h = X[1][0] - X[0][0];
    g[0][0] = -1.0 / h; g[1][0] = 1.0 / h; vol = fabs(h);
  } else {
    double T[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int d = 0; d < 3; ++d) T[r][d] = X[r + 1][d] - X[0][d];
    const double c00 = T[1][1] * T[2][2] - T[1][2] * T[2][1];
    const double c01 = T[1][2] * T[2][0] - T[1][0] * T[2][2];
    const double c02 = T[1][0] * T[2][1] - T[1][1] * T[2][0];
    const double det = T[0][0] * c00 + T[0][1] * c01 + T[0][2] * c02;
    const double id = 1.0 / det;
    // columns of T^{-1} are grad phi_1..3
    g[1][0] = c00 * id; g[1][1] = c01 * id; g[1][2] = c02 * id;
    g[2][0] = (T[0][2] * T[2][1] - T[0][1] * T[2][2]) * id;
    g[2][1] = (T[0][0] * T[2][2] - T[0][2] * T[2][0]) * id;
    g[2][2] = (T[0][1] * T[2][0] - T[0][0] * T[2][1]) * id;
    g[3][0] = (T[0][1] * T[1][2] - T[0][2] * T[1][1]) * id;
    g[3][1] = (T[0][2] * T[1][0] - T[0][0] * T[1][2]) * id;
    g[3][2] = (T[0][0] * T[1][1] - T[0][1] * T[1][0]) * id;
#pragma unroll
    for (int d = 0; d < 3; ++d) g[0][d] = -(g[1][d] + g[2][d] + g[3][d]);
    vol = fabs(det) * (1.0 / 6.0);
  }
  double gg[NN][NN];
#pragma unroll
  for (int a = 0; a < NN; ++a)
#pragma unroll
    for (int b = 0; b < NN; ++b) {
      double s = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) s += g[a][d] * g[b][d];
      gg[a][b] = s;
    }
  double gradp[DIM], G[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    double sp = 0.0, sg = 0.0;
#pragma unroll
    for (int a = 0; a < NN; ++a) {
      sp += U[a][NS] * g[a][d];
      double au = 0.0;
#pragma unroll
      for (int j = 0; j < NS; ++j) au += m.a[j] * U[a][j];
      sg += au * g[a][d];
    }
    gradp[d] = sp; G[d] = sg;
  }
  double gp[NN], Gg[NN];
#pragma unroll
  for (int a = 0; a < NN; ++a) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) { s1 += gradp[d] * g[a][d]; s2 += G[d] * g[a][d]; }
    gp[a] = s1; Gg[a] = s2;
  }
  double usum[NS], ubar[NS], epsbar = m.eps0;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < NN; ++a) s += U[a][j];
    usum[j] = s; ubar[j] = s * (1.0 / NN); epsbar += m.epsc[j] * ubar[j];
  }
  // steric moments
  double If[NS], Ij[NS], Bq[NN], Cq[NS][NN];
#pragma unroll
  for (int j = 0; j < NS; ++j) { If[j] = 0.0; Ij[j] = 0.0;
#pragma unroll
    for (int b = 0; b < NN; ++b) Cq[j][b] = 0.0; }
#pragma unroll
  for (int b = 0; b < NN; ++b) Bq[b] = 0.0;
  bool bad = false;
  if (m.steric) {
    for (int q = 0; q < qd.nq_f; ++q) {
      double uq[NS], S = 0.0;
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < NN; ++b) s += qd.lam_f[q][b] * U[b][j];
        uq[j] = s; S += m.a[j] * s;
      }
      bad |= !(1.0 - S > 0.0);
      const double wb = qd.w_f[q] * vol / (1.0 - S);
#pragma unroll
      for (int j = 0; j < NS; ++j) If[j] += wb * uq[j];
    }
    if constexpr (WANT_J) {
      for (int q = 0; q < qd.nq_j; ++q) {
        double uq[NS], S = 0.0;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          double s = 0.0;
#pragma unroll
          for (int b = 0; b < NN; ++b) s += qd.lam_j[q][b] * U[b][j];
          uq[j] = s; S += m.a[j] * s;
        }
        bad |= !(1.0 - S > 0.0);
        const double beta = 1.0 / (1.0 - S);
        const double wb = qd.w_j[q] * vol * beta;
#pragma unroll
        for (int b = 0; b < NN; ++b) Bq[b] += wb * qd.lam_j[q][b];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          Ij[j] += wb * uq[j];
          const double wbb = wb * beta * uq[j];
#pragma unroll
          for (int b = 0; b < NN; ++b) Cq[j][b] += wbb * qd.lam_j[q][b];
        }
      }
    }
  }
  if (bad) atomicOr(c.status, 1);

  // ---- element residual ------------------------------------------------------------------------
  double* ef = c.EF + (size_t)e * L::EF_STRIDE;
  // bilinear monomials int u_x u_y phi_a = |K| kappa (XY + x_a Y + X y_a + D + 2 x_a y_a)
  double mono[GMPNP_MAX_BILINEAR][NN];
  for (int t = 0; t < m.n_bilinear; ++t) {
    const int bj = m.bil_j[t], bk = m.bil_k[t];
    double xs[NN], ys[NN], Xs = 0.0, Ys = 0.0, D = 0.0;
#pragma unroll
    for (int a = 0; a < NN; ++a) {
      double xv = 0.0, yv = 0.0;
#pragma unroll
      for (int j = 0; j < NS; ++j) { xv = (j == bj) ? U[a][j] : xv; yv = (j == bk) ? U[a][j] : yv; }
      xs[a] = xv; ys[a] = yv; Xs += xv; Ys += yv; D += xv * yv;
    }
#pragma unroll
    for (int a = 0; a < NN; ++a)
      mono[t][a] = vol * L::KAPPA * (Xs * Ys + xs[a] * Ys + Xs * ys[a] + D + 2.0 * xs[a] * ys[a]);
  }
#pragma unroll
  for (int a = 0; a < NN; ++a) {
    double fp = -epsbar * vol * gp[a];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      // M-weighted nodal sums: sum_b M_ab w_b = |K| MDEN (sum_b w_b + w_a)
      double du_sum = 0.0;
#pragma unroll
      for (int b = 0; b < NN; ++b) du_sum += U[b][i] - c.un[(size_t)nd[b] * NF + i];
      const double du_a = U[a][i] - c.un[(size_t)nd[a] * NF + i];
      double f = m.inv_dt * vol * L::MDEN * (du_sum + du_a);
      double ku = 0.0;
#pragma unroll
      for (int b = 0; b < NN; ++b) ku += gg[a][b] * U[b][i];
      f += vol * ku;
      f += m.z[i] * vol * ubar[i] * gp[a];
      f += m.rc0[i] * vol * (1.0 / NN);
#pragma unroll
      for (int j = 0; j < NS; ++j) f += m.rc1[i][j] * (vol * L::MDEN * (usum[j] + U[a][j]));
      for (int t = 0; t < m.n_bilinear; ++t) f += m.rc2[i][t] * mono[t][a];
      f += If[i] * Gg[a];
      ef[a * NF + i] = f;
      fp += m.qzb[i] * (vol * L::MDEN * (usum[i] + U[a][i]));
    }
    ef[a * NF + NS] = fp;
  }
  if constexpr (WANT_J) {
    double* ej = c.EJ + (size_t)e * L::EJ_STRIDE;
    ej[L::O_VOL] = vol;
#pragma unroll
    for (int a = 0; a < NN; ++a) {
#pragma unroll
      for (int b = 0; b < NN; ++b) ej[L::O_GG + a * NN + b] = gg[a][b];
      ej[L::O_GP + a] = gp[a]; ej[L::O_GG_A + a] = Gg[a]; ej[L::O_B + a] = Bq[a];
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      ej[L::O_UBAR + j] = ubar[j]; ej[L::O_IJ + j] = Ij[j];
#pragma unroll
      for (int b = 0; b < NN; ++b) ej[L::O_C + j * NN + b] = Cq[j][b];
    }
    ej[L::O_EPS] = epsbar;
#pragma unroll
    for (int a = 0; a < NN; ++a)
#pragma unroll
      for (int j = 0; j < NS; ++j) ej[L::O_U + a * NS + j] = U[a][j];
  }
}

// ---------------------------------------------------------------------------------------------
// Residual gather: one lane per dof; b = sum of incident element rows (+ boundary terms), Dirichlet
// rows b = x - g ([3P] DirichletBC.apply(b, x)); per-workgroup partial of ||b||^2.
// ---------------------------------------------------------------------------------------------
template <int DIM, int NF>
__global__ __launch_bounds__(kVecBlock) void k_res_gather(const Ctx c) {
  using L = Lay<DIM, NF>;
  __shared__ double lds[4];
  const int r = blockIdx.x * kVecBlock + threadIdx.x;
  double val = 0.0;
  if (r < c.ndof) {
    if (c.bcflag[r]) {
      val = c.u[r] - c.bcval[r];
    } else {
      const int I = r / NF, i = r - I * NF;
      double s = c.bndF[r];
      for (int k = c.n2e_ptr[I]; k < c.n2e_ptr[I + 1]; ++k) {
        const int pk = c.n2e[k];
        const int e = pk / L::NN, a = pk - e * L::NN;
        s += c.EF[(size_t)e * L::EF_STRIDE + a * NF + i];
      }
      for (int k = c.robF_ptr[r]; k < c.robF_ptr[r + 1]; ++k) s += c.rob_val[k] * c.u[c.rob_col[k]];
      val = s;
    }
    c.F[r] = val;
  }
  double v[1] = {val * val};
  block_sum<1>(v, lds);
  if (threadIdx.x == 0) c.part_f[blockIdx.x] = v[0];
}

// ---------------------------------------------------------------------------------------------
// Jacobian gather: one wave per (slice, kpos); lane = (block row in slice, scalar row i) computes the
// NF entries of its row of block (I, cols[kpos]) by summing the element contributions in element
// order and stores them at stride 64 doubles (each store instruction writes 63 contiguous doubles).
// ---------------------------------------------------------------------------------------------
template <int DIM, int NF>
__global__ __launch_bounds__(kVecBlock) void k_jac_gather(const Ctx c) {
  using L = Lay<DIM, NF>;
  constexpr int NS = L::NS, NN = L::NN, S = L::S;
  const int wave = (blockIdx.x * kVecBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= c.n_work) return;
  const int s = c.wl_slice[wave], kpos = c.wl_kpos[wave];
  const int Iloc = lane / NF, i = lane - Iloc * NF;
  if (Iloc >= S) return;
  const int I = s * S + Iloc;
  if (I >= c.nv) return;
  const int k = c.sell_blk[(size_t)(c.slice_colbase[s] + kpos) * kSlicePad + Iloc];
  if (k < 0) return;  // padding stays zero (set at create)
  const int J = c.cols[k];
  double* out = c.vals + c.slice_off[s] + (size_t)kpos * NF * kWave + lane;
  if (c.bcflag[I * NF + i]) {  // [3P] DirichletBC.apply(A): identity row
#pragma unroll
    for (int j = 0; j < NF; ++j) out[j * kWave] = (J == I && j == i) ? 1.0 : 0.0;
    return;
  }
  const gmpnp_model_t& m = *c.model;
  const bool isp = (i == NS);
  const int is = isp ? 0 : i;
  const double zi = m.z[is], inv_dt = m.inv_dt;
  double rc1i[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) rc1i[j] = m.rc1[is][j];
  double acc[NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) acc[j] = 0.0;

#pragma unroll 4
  for (int q = c.cptr[k]; q < c.cptr[k + 1]; ++q) {
    const int pk = c.contrib[q];
    const int e = pk >> 4, a = (pk >> 2) & 3, b = pk & 3;
    const double* ej = c.EJ + (size_t)e * L::EJ_STRIDE;
    const double vol = ej[L::O_VOL], ggab = ej[L::O_GG + a * NN + b], gpa = ej[L::O_GP + a];
    const double Mab = vol * L::MDEN * (a == b ? 2.0 : 1.0), Kab = vol * ggab;
    if (!isp) {
      const double Gga = ej[L::O_GG_A + a];
      const double ster = Gga * ej[L::O_C + is * NN + b] + ej[L::O_IJ + is] * ggab;
      const double dg = inv_dt * Mab + Kab + zi * vol * (1.0 / NN) * gpa + Gga * ej[L::O_B + b];
#pragma unroll
      for (int j = 0; j < NS; ++j) acc[j] += m.a[j] * ster + rc1i[j] * Mab + (j == is ? dg : 0.0);
      for (int t = 0; t < m.n_bilinear; ++t) {
        const double c2 = m.rc2[is][t];
        if (c2 != 0.0) {
          const int bj = m.bil_j[t], bk = m.bil_k[t];
          const double xa = ej[L::O_U + a * NS + bj], xb = ej[L::O_U + b * NS + bj];
          const double ya = ej[L::O_U + a * NS + bk], yb = ej[L::O_U + b * NS + bk];
          const double Xs = NN * ej[L::O_UBAR + bj], Ys = NN * ej[L::O_UBAR + bk];
          const double w = c2 * vol * L::KAPPA;
          const double dj = w * (Ys + ya + yb + (a == b ? Ys + 2.0 * ya : 0.0));  // d/d u_{bj,b}
          const double dk = w * (Xs + xa + xb + (a == b ? Xs + 2.0 * xa : 0.0));  // d/d u_{bk,b}
#pragma unroll
          for (int j = 0; j < NS; ++j) acc[j] += (j == bj ? dj : 0.0) + (j == bk ? dk : 0.0);
        }
      }
      acc[NS] += zi * ej[L::O_UBAR + is] * Kab;
    } else {
      const double kpa = vol * gpa * (1.0 / NN);
#pragma unroll
      for (int j = 0; j < NS; ++j) acc[j] += -m.epsc[j] * kpa + m.qzb[j] * Mab;
      acc[NS] += -ej[L::O_EPS] * Kab;
    }
  }
#pragma unroll
  for (int j = 0; j < NF; ++j) out[j * kWave] = acc[j];
}

// Robin (exit) mass entries, one lane per pre-merged entry (unique addresses: no atomics).
__global__ void k_robin_add(const Ctx c) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= c.n_robin) return;
  if (c.bcflag[c.rob_row[t]]) return;
  c.vals[c.rob_addr[t]] += c.rob_val[t];
}

// ---------------------------------------------------------------------------------------------
// Node-block Jacobi: invert the NF x NF diagonal blocks (Gauss-Jordan, partial pivoting, LDS-resident).
// ---------------------------------------------------------------------------------------------
template <int NF>
__global__ __launch_bounds__(64) void k_block_inverse(const Ctx c) {
  constexpr int S = kWave / NF;
  __shared__ double A[NF * NF][64];
  const int I = blockIdx.x * 64 + threadIdx.x, t = threadIdx.x;
  if (I >= c.nv) return;
  const int s = I / S, Iloc = I - s * S;
  const double* base = c.vals + c.slice_off[s] + Iloc * NF;  // the diagonal block is SELL position 0
  for (int i = 0; i < NF; ++i)
    for (int j = 0; j < NF; ++j) A[i * NF + j][t] = base[(size_t)j * kWave + i];
  int piv[NF];
  bool sing = false;
  for (int k = 0; k < NF; ++k) {
    int p = k; double best = fabs(A[k * NF + k][t]);
    for (int r = k + 1; r < NF; ++r) { const double v = fabs(A[r * NF + k][t]); if (v > best) { best = v; p = r; } }
    piv[k] = p;
    if (!(best > 0.0)) { sing = true; break; }
    if (p != k)
      for (int j = 0; j < NF; ++j) { const double tmp = A[k * NF + j][t]; A[k * NF + j][t] = A[p * NF + j][t]; A[p * NF + j][t] = tmp; }
    const double ip = 1.0 / A[k * NF + k][t];
    A[k * NF + k][t] = 1.0;
    for (int j = 0; j < NF; ++j) A[k * NF + j][t] *= ip;
    for (int r = 0; r < NF; ++r) {
      if (r == k) continue;
      const double f = A[r * NF + k][t];
      A[r * NF + k][t] = 0.0;
      for (int j = 0; j < NF; ++j) A[r * NF + j][t] -= f * A[k * NF + j][t];
    }
  }
  if (sing) { atomicOr(c.status, 2); return; }
  for (int k = NF - 1; k >= 0; --k)
    if (piv[k] != k)
      for (int r = 0; r < NF; ++r) { const double tmp = A[r * NF + k][t]; A[r * NF + k][t] = A[r * NF + piv[k]][t]; A[r * NF + piv[k]][t] = tmp; }
  double* o = c.Dinv + (size_t)I * NF * NF;
  for (int q = 0; q < NF * NF; ++q) o[q] = A[q][t];
}

// ---------------------------------------------------------------------------------------------
// Coarse operator Ac = P^T A P in two deterministic stages.
//   stage 1 (k_coarse_rows): AP[r][slot][j] = sum over the row's blocks whose column node lies in aggregate slot
//   stage 2 (k_coarse_sum) : Ac[g*NF+i][h*NF+j] = sum over nodes I in g of AP[(I,i)][slot(h)][j]
// ---------------------------------------------------------------------------------------------
template <int NF>
__global__ __launch_bounds__(64) void k_coarse_rows(const Ctx c) {
  constexpr int S = kWave / NF;
  const int s = blockIdx.x, lane = threadIdx.x;
  const int Iloc = lane / NF, i = lane - Iloc * NF;
  if (Iloc >= S) return;
  const int I = s * S + Iloc;
  if (I >= c.nv) return;
  double acc[kMaxRowAggs][NF];
#pragma unroll
  for (int q = 0; q < kMaxRowAggs; ++q)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[q][j] = 0.0;
  const int cb = c.slice_colbase[s], mx = c.slice_colbase[s + 1] - cb;
  const double* base = c.vals + c.slice_off[s] + lane;
  for (int kp = 0; kp < mx; ++kp) {
    const int slot = c.sell_aggslot[(size_t)(cb + kp) * kSlicePad + Iloc];
    if (slot == 255) continue;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const double v = base[(size_t)(kp * NF + j) * kWave];
#pragma unroll
      for (int q = 0; q < kMaxRowAggs; ++q) acc[q][j] += (q == slot) ? v : 0.0;
    }
  }
  double* o = c.AP + (size_t)(I * NF + i) * kMaxRowAggs * NF;
#pragma unroll
  for (int q = 0; q < kMaxRowAggs; ++q)
#pragma unroll
    for (int j = 0; j < NF; ++j) o[q * NF + j] = acc[q][j];
}

template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_coarse_sum(const Ctx c) {
  // workgroup (g, chunk): partial sums over one chunk of the nodes of aggregate g
  const int g = blockIdx.x / kCoarseChunks, ch = blockIdx.x - g * kCoarseChunks, n = c.ncoarse;
  const int a0 = c.agg_start[g], len = c.agg_start[g + 1] - a0;
  const int I0 = a0 + (int)((int64_t)len * ch / kCoarseChunks), I1 = a0 + (int)((int64_t)len * (ch + 1) / kCoarseChunks);
  double* out = c.AcPart + ((size_t)ch * n + (size_t)g * NF) * n;
  for (int idx = threadIdx.x; idx < NF * n; idx += kVecBlock) {
    const int i = idx / n, col = idx - i * n, h = col / NF, j = col - h * NF;
    double s = 0.0;
    for (int I = I0; I < I1; ++I) {
      const int32_t* ra = c.row_aggs + (size_t)I * kMaxRowAggs;
#pragma unroll
      for (int q = 0; q < kMaxRowAggs; ++q)
        if (ra[q] == h) s += c.AP[((size_t)(I * NF + i) * kMaxRowAggs + q) * NF + j];
    }
    out[(size_t)i * n + col] = s;
  }
}

__global__ __launch_bounds__(kVecBlock) void k_coarse_reduce(const Ctx c) {
  const int n2 = c.ncoarse * c.ncoarse, q = blockIdx.x * kVecBlock + threadIdx.x;
  if (q >= n2) return;
  double s = 0.0;
  for (int ch = 0; ch < kCoarseChunks; ++ch) s += c.AcPart[(size_t)ch * n2 + q];
  c.Ac[q] = s;
}

// In-place BLOCK Gauss-Jordan inverse of the coarse operator (block = one aggregate, NF x NF) by one
// 512-thread workgroup with the whole matrix resident in LDS.  Pivoting happens inside the diagonal
// block only (its inverse is formed by NF lanes with partial pivoting); the off-diagonal update is a
// rank-NF update done in 3x3 register tiles.
template <int NF>
__global__ __launch_bounds__(512) void k_coarse_invert(const Ctx c) {
  extern __shared__ double sm[];
  const int n = c.ncoarse, nblk = n / NF, t = threadIdx.x, nt = blockDim.x;
  double* A = sm;                 // n*n
  double* colK = A + n * n;       // n*NF : old column block K
  double* D = colK + n * NF;      // NF*(2*NF) augmented [A_KK | I] -> [I | inv]
  __shared__ int sing;
  if (t == 0) sing = 0;
  for (int q = t; q < n * n; q += nt) A[q] = c.Ac[q];
  __syncthreads();
  const int ntile = (n + 2) / 3;
  for (int K = 0; K < nblk; ++K) {
    const int k0 = K * NF;
    // (1) D = inverse of the diagonal block: lanes 0..NF-1 of wave 0 own one row each
    if (t < 64) {
      const int r = t;
      double row[2 * NF];
      if (r < NF) {
#pragma unroll
        for (int j = 0; j < NF; ++j) { row[j] = A[(k0 + r) * n + k0 + j]; row[NF + j] = (j == r) ? 1.0 : 0.0; }
      }
      bool bad = false;
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        // pivot row among lanes k..NF-1
        double v = (r < NF && r >= k) ? fabs(row[k]) : -1.0; int idx = r;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          const double ov = __shfl_xor(v, o, 16); const int oi = __shfl_xor(idx, o, 16);
          if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        bad |= !(v > 0.0);
        // rows k and idx swap places (lane k takes the pivot row, lane idx the old row k), then scale / eliminate
        const double ip = 1.0 / __shfl(row[k], idx, 16);
        const double oldk_k = __shfl(row[k], k, 16);
        const double f = ((r == idx) ? oldk_k : row[k]) * ip;
#pragma unroll
        for (int j = 0; j < 2 * NF; ++j) {
          const double from_p = __shfl(row[j], idx, 16), from_k = __shfl(row[j], k, 16);
          const double mine = (r == idx) ? from_k : row[j];
          row[j] = (r == k) ? from_p * ip : mine - f * from_p;
        }
      }
      if (r < NF) {
#pragma unroll
        for (int j = 0; j < NF; ++j) D[r * NF + j] = row[NF + j];
      }
      if (bad && t == 0) sing = 1;
    }
    // (2a) save the old column block K
    for (int q = t; q < n * NF; q += nt) { const int r = q / NF, m = q - r * NF; colK[q] = A[r * n + k0 + m]; }
    __syncthreads();
    if (sing) break;
    // (2b) row block K: A[K,K] = D ; A[K,c] = D * A[K,c]   (column-wise, each thread one column)
    for (int cc = t; cc < n; cc += nt) {
      double colv[NF];
      const bool inK = (cc >= k0 && cc < k0 + NF);
#pragma unroll
      for (int m = 0; m < NF; ++m) colv[m] = A[(k0 + m) * n + cc];
#pragma unroll
      for (int r = 0; r < NF; ++r) {
        double sacc = 0.0;
        if (inK) sacc = D[r * NF + (cc - k0)];
        else
#pragma unroll
          for (int m = 0; m < NF; ++m) sacc += D[r * NF + m] * colv[m];
        A[(k0 + r) * n + cc] = sacc;
      }
    }
    __syncthreads();
    // (3) other row blocks: A[r][c] = (c in K ? 0 : A[r][c]) - sum_m colK[r][m] * A[K*NF+m][c]
    for (int tile = t; tile < ntile * ntile; tile += nt) {
      const int tr = tile / ntile, tc = tile - tr * ntile;
      const int r0 = 3 * tr, c0 = 3 * tc;
      double acc[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) acc[a][b2] = 0.0;
#pragma unroll
      for (int m = 0; m < NF; ++m) {
        double cv[3], wv[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) cv[a] = (r0 + a < n) ? colK[(r0 + a) * NF + m] : 0.0;
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) wv[b2] = (c0 + b2 < n) ? A[(k0 + m) * n + c0 + b2] : 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) acc[a][b2] += cv[a] * wv[b2];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int r = r0 + a;
        if (r >= n || (r >= k0 && r < k0 + NF)) continue;
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) {
          const int cc = c0 + b2;
          if (cc >= n) continue;
          const bool inK = (cc >= k0 && cc < k0 + NF);
          A[r * n + cc] = (inK ? 0.0 : A[r * n + cc]) - acc[a][b2];
        }
      }
    }
    __syncthreads();
  }
  if (sing) { if (t == 0) atomicOr(c.status, 4); return; }
  for (int q = t; q < n * n; q += nt) c.AciT[q] = A[q];  // row-major inverse (name kept: see k_coarse)
}

// ---------------------------------------------------------------------------------------------
// Coarse solve yc = Aci pc, one wave per coarse row (4 rows per workgroup).  pc = fixed-order sum of the
// restriction partials pc_part[coarse dof][slot] (slots of absent workgroups stay zero).  The matrix row and
// the partials are requested together, so the kernel is one memory round trip plus a wave reduction.
// ---------------------------------------------------------------------------------------------
template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_coarse(const Ctx c, int use_coarse) {
  __shared__ double pc[kMaxCoarse];
  if (c.scal->done) return;
  const int n = c.ncoarse, t = threadIdx.x;
  const int row = blockIdx.x * 4 + (t >> 6), lane = t & 63;
  if (!use_coarse) { if (lane == 0 && row < n) c.yc[row] = 0.0; return; }
  double a[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { const int cc = lane + 64 * k; a[k] = (row < n && cc < n) ? c.AciT[(size_t)row * n + cc] : 0.0; }
  if (t < n) {
    const double* pp = c.pc_part + (size_t)t * c.vw_slots;
    double s = 0.0;
#pragma unroll 16
    for (int k = 0; k < c.vw_slots; ++k) s += pp[k];
    pc[t] = s;
  }
  __syncthreads();
  if (row >= n) return;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) { const int cc = lane + 64 * k; if (cc < n) s += a[k] * pc[cc]; }
  s = wave_sum(s);
  if (lane == 0) c.yc[row] = s;
}

// ---------------------------------------------------------------------------------------------
// SELL block SpMV: out = A (x + P yc).  One workgroup (NW waves) per slice, the waves split the block columns;
// lane = (block row in slice, scalar row).  Every value load is a 504-byte contiguous wave read.
// MODE 0: plain.  MODE 1: part_a = (rhat, out).  MODE 2: part_b = (out,s) (out,out) (rhat,s) (rhat,out).
// ---------------------------------------------------------------------------------------------
template <int NF, int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k_spmv(const Ctx c, const double* __restrict__ x, double* __restrict__ out) {
  constexpr int S = kWave / NF;
  __shared__ double red[NW][64];
  if (MODE != 0 && c.scal->done) return;
  const int s = blockIdx.x, t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int Iloc = lane / NF, i = lane - Iloc * NF;
  const int I = s * S + Iloc;
  const bool active = (Iloc < S) && (I < c.nv);
  const int cb = c.slice_colbase[s], mx = c.slice_colbase[s + 1] - cb;
  const double* base = c.vals + c.slice_off[s] + lane;
  double rh = 0.0, sv = 0.0;  // operands of the fused dot products: requested early
  if (MODE != 0 && w == 0 && active) { rh = c.krhat[I * NF + i]; if (MODE == 2) sv = c.ks[I * NF + i]; }
  double acc = 0.0;
  if (active) {
#pragma unroll 2
    for (int kp = w; kp < mx; kp += NW) {
      const int pk = c.sell_cols[(size_t)(cb + kp) * kSlicePad + Iloc];
      const double* xv = x + (size_t)(pk & 0xFFFFFF) * NF;
      const double* yv = c.yc + (pk >> 24) * NF;
      const double* av = base + (size_t)kp * NF * kWave;
#pragma unroll
      for (int j = 0; j < NF; ++j) acc += av[(size_t)j * kWave] * (xv[j] + yv[j]);
    }
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w != 0) return;
  double tot = 0.0;
#pragma unroll
  for (int q = 0; q < NW; ++q) tot += red[q][lane];
  const int r = I * NF + i;
  if (active) out[r] = tot;
  if (MODE == 1) {
    const double d = wave_sum(active ? rh * tot : 0.0);
    if (lane == 0) c.part_a[s] = d;
  } else if (MODE == 2) {
    const double tv = active ? tot : 0.0;
    const double d0 = wave_sum(tv * sv), d1 = wave_sum(tv * tv), d2 = wave_sum(rh * sv), d3 = wave_sum(rh * tv);
    if (lane == 0) {
      c.part_b[s] = d0; c.part_b[c.nslices + s] = d1; c.part_b[2 * c.nslices + s] = d2; c.part_b[3 * c.nslices + s] = d3;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// BiCGStab vector kernels.  Workgroup = whole nodes of ONE aggregate (so the restriction partial is a
// per-workgroup sum), thread = dof.  Right preconditioning: the Krylov space lives on y, x = M^{-1} y.
//
// k_vec1: (iteration > 0) omega = (t,s)/(t,t); y += alpha p + omega s; r = s - omega t;
//         rho' = (rhat,s) - omega (rhat,t); beta = (rho'/rho)(alpha/omega); p = r + beta (p - omega v)
//         (iteration 0)  p = r
//         then q = Dinv p, restriction partial of p, partial of (r,r).
// k_vec2: convergence test on (r,r); alpha = rho'/(rhat,v); s = r - alpha v; q = Dinv s; restriction of s.
// ---------------------------------------------------------------------------------------------
template <int NF>
__device__ inline void dinv_restrict(const Ctx& c, int wg, int n0, int nnodes, int tid, bool valid, double val,
                                     const double (&drow)[NF], double* lv /* [kVecBlock] */) {
  lv[tid] = valid ? val : 0.0;
  __syncthreads();
  if (valid) {
    const int nl = tid / NF;
    double q = 0.0;
#pragma unroll
    for (int mI = 0; mI < NF; ++mI) q += drow[mI] * lv[nl * NF + mI];
    c.kq[n0 * NF + tid] = q;
  }
  if (tid < NF) {
    double s = 0.0;
    for (int nl = 0; nl < nnodes; ++nl) s += lv[nl * NF + tid];
    const int g = c.vw_agg[wg];
    c.pc_part[(size_t)(g * NF + tid) * c.vw_slots + (wg - c.agg_vw_ptr[g])] = s;
  }
}

template <int NF>
__device__ inline void load_dinv_row(const Ctx& c, int r, bool valid, double (&drow)[NF]) {
#pragma unroll
  for (int mI = 0; mI < NF; ++mI) drow[mI] = valid ? c.Dinv[(size_t)r * NF + mI] : 0.0;
}

template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_vec1(const Ctx c) {
  __shared__ double lv[kVecBlock];
  __shared__ double lred[16];
  KrylovScalars* sc = c.scal;
  if (sc->done) return;
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int n0 = c.vw_node0[wg], nnodes = c.vw_node1[wg] - n0;
  const bool valid = tid < nnodes * NF;
  const int r = n0 * NF + tid;
  const int iters = sc->iters;
  const bool first = (iters == 0);
  // every operand is requested before the partial sums are reduced (independent round trips overlap)
  double drow[NF];
  load_dinv_row<NF>(c, r, valid, drow);
  double sv = 0.0, tv = 0.0, pold = 0.0, vv = 0.0, yv = 0.0, r0 = 0.0;
  if (valid) {
    if (first) r0 = c.kr[r];
    else { sv = c.ks[r]; tv = c.kt[r]; pold = c.kp[r]; vv = c.kv[r]; yv = c.ky[r]; }
  }
  double pv = 0.0, rn = 0.0, rho_next;
  if (first) {
    rho_next = sc->rho;
    rn = r0; pv = r0;
  } else {
    const double alpha = sc->alpha, rho = sc->rho;
    double tot[4];
    sum_partials<4>(c.part_b, c.nslices, c.nslices, tot, lred);
    const double ts = tot[0], tt = tot[1], rs = tot[2], rt = tot[3];
    const double omega = ts / tt;
    rho_next = rs - omega * rt;
    const double beta = (rho_next / rho) * (alpha / omega);
    if (valid) {
      c.ky[r] = yv + alpha * pold + omega * sv;
      rn = sv - omega * tv;
      c.kr[r] = rn;
      pv = rn + beta * (pold - omega * vv);
    }
  }
  if (valid) c.kp[r] = pv;
  dinv_restrict<NF>(c, wg, n0, nnodes, tid, valid, pv, drow, lv);
  double v[1] = {rn * rn};
  block_sum<1>(v, lred);
  if (tid == 0) {
    c.part_rr[wg] = v[0];
    if (wg == 0) { sc->rho_next = rho_next; sc->it_cur = iters; }
  }
}

template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_vec2(const Ctx c) {
  __shared__ double lv[kVecBlock];
  __shared__ double lred[16];
  KrylovScalars* sc = c.scal;
  if (sc->done) return;
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int n0 = c.vw_node0[wg], nnodes = c.vw_node1[wg] - n0;
  const bool valid = tid < nnodes * NF;
  const int r = n0 * NF + tid;
  // vec2 only reads scalars that vec1 (the previous launch) wrote and only writes scalars that vec1 reads
  const double rho_next = sc->rho_next, tol = sc->tol;
  const int iters = sc->it_cur, max_iters = sc->max_iters;
  double drow[NF];
  load_dinv_row<NF>(c, r, valid, drow);
  const double rv_ = valid ? c.kr[r] : 0.0, vv = valid ? c.kv[r] : 0.0;
  double rr[1], rv[1];
  sum_partials<1>(c.part_rr, c.n_vecwg, c.n_vecwg, rr, lred);
  sum_partials<1>(c.part_a, c.nslices, c.nslices, rv, lred);
  int done = 0;
  if (!(rr[0] == rr[0]) || !(rv[0] == rv[0])) done = 3;                  // NaN
  else if (sqrt(rr[0]) <= tol) done = 1;
  else if (iters >= max_iters) done = 2;
  else if (rv[0] == 0.0 || rho_next == 0.0) done = 3;                    // breakdown
  if (done) {
    if (wg == 0 && tid == 0) { sc->rr = rr[0]; sc->done = done; }
    return;
  }
  const double alpha = rho_next / rv[0];
  double sv = 0.0;
  if (valid) { sv = rv_ - alpha * vv; c.ks[r] = sv; }
  dinv_restrict<NF>(c, wg, n0, nnodes, tid, valid, sv, drow, lv);
  if (wg == 0 && tid == 0) { sc->alpha = alpha; sc->rho = rho_next; sc->rr = rr[0]; sc->iters = iters + 1; }
}

// x = M^{-1} y = Dinv y + P Aci P^T y, in three steps: k_vec_final (q = Dinv y, restrict y), k_coarse, k_apply.
template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_vec_final(const Ctx c) {
  __shared__ double lv[kVecBlock];
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int n0 = c.vw_node0[wg], nnodes = c.vw_node1[wg] - n0;
  const bool valid = tid < nnodes * NF;
  double drow[NF];
  load_dinv_row<NF>(c, n0 * NF + tid, valid, drow);
  dinv_restrict<NF>(c, wg, n0, nnodes, tid, valid, valid ? c.ky[n0 * NF + tid] : 0.0, drow, lv);
}

// dst[r] = scale_dst * dst[r] + scale_x * (q[r] + yc[agg]);  Newton update: u -= omega dx  (scale_dst 1, scale_x -omega)
template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_apply(const Ctx c, double* __restrict__ dst, double scale_dst, double scale_x) {
  const int r = blockIdx.x * kVecBlock + threadIdx.x;
  if (r >= c.ndof) return;
  const int I = r / NF, f = r - I * NF;
  const double x = c.kq[r] + c.yc[c.agg[I] * NF + f];
  dst[r] = (scale_dst == 0.0 ? 0.0 : scale_dst * dst[r]) + scale_x * x;
}

__global__ void k_copy2(double* __restrict__ a, double* __restrict__ b, const double* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double v = src[i]; a[i] = v; if (b) b[i] = v; }
}

__global__ void k_fill(double* __restrict__ a, double v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}


// =================================================================================================
// Direct solver for 1D meshes: block cyclic reduction of the block-tridiagonal Jacobian (NF x NF blocks).
// Replaces the reference's default sparse LU of the 1D script (1D/MPNP_CO2ER_EDL.py:357-364, UMFPACK) where a
// Jacobi-type Krylov method is hopeless (q ~ 1e9, 0.1 nm cells next to 10 nm cells).
//
// Level l holds n_l rows in SoA form ([entry][row]); level l+1 = the even rows of level l after eliminating
// the odd ones.  One lane per even row does both neighbour eliminations (Gauss-Jordan with partial pivoting
// inside the NF x NF block, scratch in LDS) and keeps D_j^{-1}[L_j U_j b_j] of its right odd neighbour for the
// back substitution.  ~log2(n) launches down and up; no atomics, deterministic.
// =================================================================================================
struct TriLevel {
  int n;
  double *L, *D, *U, *b;     // [NF*NF][n] x3, [NF][n]
  double *Li, *Ui, *bi;      // D^{-1}L, D^{-1}U, D^{-1}b of the odd rows
  double* x;                 // [NF][n]
};

// Solve Dm X = [Lm | Um | bm] for row j of level lv; result left in W[(r*(3*NF+1) + NF + c)*64 + t], c in [0, 2NF].
template <int NF>
__device__ inline bool tri_inv_apply(const TriLevel& lv, int j, double* W, int t) {
  constexpr int NC = 3 * NF + 1;
  for (int r = 0; r < NF; ++r) {
    for (int cI = 0; cI < NF; ++cI) {
      W[(r * NC + cI) * 64 + t] = lv.D[(size_t)(r * NF + cI) * lv.n + j];
      W[(r * NC + NF + cI) * 64 + t] = lv.L[(size_t)(r * NF + cI) * lv.n + j];
      W[(r * NC + 2 * NF + cI) * 64 + t] = lv.U[(size_t)(r * NF + cI) * lv.n + j];
    }
    W[(r * NC + 3 * NF) * 64 + t] = lv.b[(size_t)r * lv.n + j];
  }
  for (int k = 0; k < NF; ++k) {
    int p = k; double best = fabs(W[(k * NC + k) * 64 + t]);
    for (int r = k + 1; r < NF; ++r) { const double v = fabs(W[(r * NC + k) * 64 + t]); if (v > best) { best = v; p = r; } }
    if (!(best > 0.0)) return false;
    if (p != k)
      for (int cI = k; cI < NC; ++cI) { const double tmp = W[(k * NC + cI) * 64 + t]; W[(k * NC + cI) * 64 + t] = W[(p * NC + cI) * 64 + t]; W[(p * NC + cI) * 64 + t] = tmp; }
    const double ip = 1.0 / W[(k * NC + k) * 64 + t];
    for (int cI = k; cI < NC; ++cI) W[(k * NC + cI) * 64 + t] *= ip;
    for (int r = 0; r < NF; ++r) {
      if (r == k) continue;
      const double f = W[(r * NC + k) * 64 + t];
      if (f == 0.0) continue;
      for (int cI = k; cI < NC; ++cI) W[(r * NC + cI) * 64 + t] -= f * W[(k * NC + cI) * 64 + t];
    }
  }
  return true;
}

template <int NF>
__global__ __launch_bounds__(64) void k_bcr_forward(TriLevel lo, TriLevel hi, int32_t* status) {
  constexpr int NC = 3 * NF + 1;
  extern __shared__ double W[];  // [NF*NC][64]
  const int t = threadIdx.x, ih = blockIdx.x * 64 + t;  // row of the upper level
  if (ih >= hi.n) return;
  const int i = 2 * ih;
  bool ok = true;
  // start from row i itself
  for (int e = 0; e < NF * NF; ++e) {
    hi.D[(size_t)e * hi.n + ih] = lo.D[(size_t)e * lo.n + i];
    hi.L[(size_t)e * hi.n + ih] = 0.0;
    hi.U[(size_t)e * hi.n + ih] = 0.0;
  }
  for (int r = 0; r < NF; ++r) hi.b[(size_t)r * hi.n + ih] = lo.b[(size_t)r * lo.n + i];
  for (int side = 0; side < 2; ++side) {
    const int j = side == 0 ? i - 1 : i + 1;
    if (j < 0 || j >= lo.n) continue;
    ok &= tri_inv_apply<NF>(lo, j, W, t);
    const double* C = side == 0 ? lo.L : lo.U;  // coupling of row i to row j
    for (int r = 0; r < NF; ++r) {
      double cr[NF];
#pragma unroll
      for (int mI = 0; mI < NF; ++mI) cr[mI] = C[(size_t)(r * NF + mI) * lo.n + i];
      for (int cI = 0; cI < NF; ++cI) {
        double sL = 0.0, sU = 0.0;
#pragma unroll
        for (int mI = 0; mI < NF; ++mI) { sL += cr[mI] * W[(mI * NC + NF + cI) * 64 + t]; sU += cr[mI] * W[(mI * NC + 2 * NF + cI) * 64 + t]; }
        if (side == 0) { hi.L[(size_t)(r * NF + cI) * hi.n + ih] = -sL; hi.D[(size_t)(r * NF + cI) * hi.n + ih] -= sU; }
        else { hi.D[(size_t)(r * NF + cI) * hi.n + ih] -= sL; hi.U[(size_t)(r * NF + cI) * hi.n + ih] = -sU; }
      }
      double sb = 0.0;
#pragma unroll
      for (int mI = 0; mI < NF; ++mI) sb += cr[mI] * W[(mI * NC + 3 * NF) * 64 + t];
      hi.b[(size_t)r * hi.n + ih] -= sb;
    }
    if (side == 1) {  // keep the right neighbour's solved couplings for the way back up
      for (int r = 0; r < NF; ++r) {
        for (int cI = 0; cI < NF; ++cI) {
          lo.Li[(size_t)(r * NF + cI) * lo.n + j] = W[(r * NC + NF + cI) * 64 + t];
          lo.Ui[(size_t)(r * NF + cI) * lo.n + j] = W[(r * NC + 2 * NF + cI) * 64 + t];
        }
        lo.bi[(size_t)r * lo.n + j] = W[(r * NC + 3 * NF) * 64 + t];
      }
    }
  }
  if (!ok) atomicOr(status, 2);
}

// top of the pyramid: one row, x = D^{-1} b
template <int NF>
__global__ __launch_bounds__(64) void k_bcr_top(TriLevel top, int32_t* status) {
  extern __shared__ double W[];
  constexpr int NC = 3 * NF + 1;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!tri_inv_apply<NF>(top, 0, W, 0)) { atomicOr(status, 2); return; }
  for (int r = 0; r < NF; ++r) top.x[r] = W[(r * NC + 3 * NF) * 64];
}

template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_bcr_backward(TriLevel lo, TriLevel hi) {
  const int j = blockIdx.x * kVecBlock + threadIdx.x;
  if (j >= lo.n) return;
  if ((j & 1) == 0) {
#pragma unroll
    for (int r = 0; r < NF; ++r) lo.x[(size_t)r * lo.n + j] = hi.x[(size_t)r * hi.n + (j >> 1)];
    return;
  }
  double xl[NF], xr[NF];
  const bool has_r = (j + 1 < lo.n);
#pragma unroll
  for (int r = 0; r < NF; ++r) {
    xl[r] = hi.x[(size_t)r * hi.n + ((j - 1) >> 1)];
    xr[r] = has_r ? hi.x[(size_t)r * hi.n + ((j + 1) >> 1)] : 0.0;
  }
  for (int r = 0; r < NF; ++r) {
    double s = lo.bi[(size_t)r * lo.n + j];
#pragma unroll
    for (int cI = 0; cI < NF; ++cI)
      s -= lo.Li[(size_t)(r * NF + cI) * lo.n + j] * xl[cI] + lo.Ui[(size_t)(r * NF + cI) * lo.n + j] * xr[cI];
    lo.x[(size_t)r * lo.n + j] = s;
  }
}

// SELL Jacobian + right-hand side -> level-0 SoA arrays (internal order must be the path order of the interval mesh)
template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_tri_extract(const Ctx c, TriLevel l0, const int32_t* __restrict__ tri_kpos,
                                                           const double* __restrict__ rhs) {
  constexpr int S = kWave / NF;
  const int idx = blockIdx.x * kVecBlock + threadIdx.x;
  if (idx >= c.nv * NF * NF) return;
  const int e = idx / c.nv, I = idx - e * c.nv, i = e / NF, j = e - i * NF;
  const int s = I / S, il = I - s * S;
  const double* base = c.vals + c.slice_off[s] + il * NF + i;
  const int kl = tri_kpos[I * 3], kd = tri_kpos[I * 3 + 1], kr = tri_kpos[I * 3 + 2];
  l0.L[idx] = kl >= 0 ? base[(size_t)(kl * NF + j) * kWave] : 0.0;
  l0.D[idx] = base[(size_t)(kd * NF + j) * kWave];
  l0.U[idx] = kr >= 0 ? base[(size_t)(kr * NF + j) * kWave] : 0.0;
  if (e < NF) l0.b[(size_t)e * c.nv + I] = rhs[(size_t)I * NF + e];
}

// dst[I*NF+f] = scale_dst*dst + scale_x * x[f][I]
template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_tri_apply(TriLevel l0, double* __restrict__ dst, double scale_dst, double scale_x, int ndof) {
  const int r = blockIdx.x * kVecBlock + threadIdx.x;
  if (r >= ndof) return;
  const int I = r / NF, f = r - I * NF;
  dst[r] = (scale_dst == 0.0 ? 0.0 : scale_dst * dst[r]) + scale_x * l0.x[(size_t)f * l0.n + I];
}

}  // namespace gmpnp
