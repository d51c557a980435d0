// HIP kernels of the GMPNP hot path for gfx950 (wave64, fp64 VALU; no MFMA: there is no dense contraction).
//
//   k_element      per-element P1 residual + the few scalars the exact Jacobian is built from     (a1-a4)
//   k_res_gather   race-free node gather of the residual, Dirichlet rows b = x - g, ||b||^2      (a5, a6)
//   k_jac_gather   race-free gather of the node-block Jacobian straight into SELL storage,
//                  identity Dirichlet rows                                                        (a5, a6)
//   k_block_inverse, k_scale_columns, k_coarse_rows/_sum/_reduce/_invert
//                  preconditioner set-up: node-block inverses, As = J Dinv, Galerkin coarse operator, its inverse   (a7)
//   k_coarse_a/b, k_bicg_a/b  (four launches per iteration)  or  k_half_a/b  (two: coarse workgroups inside the tile launch)
//                  fused right-preconditioned BiCGStab: scalars + coarse solve, vector updates + SELL SpMV + dots    (a7)
//   k_bcr_*, k_tri_*   1D: block cyclic reduction of the block-tridiagonal Jacobian (direct solve)                   (a7)
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
  static constexpr int O_D = O_C + NS * NN;       // [MAX_BILINEAR][2][NN][NN] d(int u_x u_y phi_a)/d u_{x,b} and /d u_{y,b}
  static constexpr int O_S = O_D + GMPNP_MAX_BILINEAR * 2 * NN * NN;  // 1D only: [NN][NF][NN][NF] SUPG element matrix (PNP + stabilisation)
  static constexpr int EJ_STRIDE = O_S + (DIM == 1 ? NN * NF * NN * NF : 0);
  static constexpr int EF_STRIDE = NN * NF;
  static constexpr double MDEN = 1.0 / ((DIM + 1) * (DIM + 2));  // M_ab = |K| (1+delta_ab) MDEN
  static constexpr double KAPPA = (DIM == 3) ? 1.0 / 120.0 : 1.0 / 24.0;  // d!/(d+3)!
};

// ---------------------------------------------------------------------------------------------
// deterministic workgroup reductions (256 threads = 4 waves)
// ---------------------------------------------------------------------------------------------
// Sum over the 64 lanes of a wave, result in every lane.  DPP row shifts / row broadcasts (gfx9 data-parallel primitives:
// plain VALU moves, no LDS crossbar) instead of six dependent ds_bpermute round trips: lanes shifted in from outside a
// row read 0, after four shifts lane 15 of each row holds the row total, row_bcast:15 / :31 carry the totals across
// rows into lane 63, which is read back into a scalar.  Fixed order: bitwise reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double wave_sum(double v) {
  v += dpp_or_zero<0x111, 0xf>(v);  // row_shr:1
  v += dpp_or_zero<0x112, 0xf>(v);  // row_shr:2
  v += dpp_or_zero<0x114, 0xf>(v);  // row_shr:4
  v += dpp_or_zero<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row: row total
  v += dpp_or_zero<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_or_zero<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63: wave total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
// Sum over aligned groups of 8 lanes; the total lands in the LAST lane of each group (lanes 7, 15, ... of the wave).
__device__ inline double group8_sum_to_last(double v) {
  v += dpp_or_zero<0x111, 0xf>(v);  // row_shr:1
  v += dpp_or_zero<0x112, 0xf>(v);  // row_shr:2
  v += dpp_or_zero<0x114, 0xf>(v);  // row_shr:4  -> lane 8k+7: lanes 8k .. 8k+7
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
// STAGED (3D meshes): a lane's record words are collected in LDS piece by piece (at most 64 words at a time) and the WAVE writes
// them out record by record, up to 512 contiguous bytes per store instruction.  With one lane per cell and direct stores every
// store instruction touches 64 records 1.6 KB apart, and a wave keeps 64 x 1.9 KB of half-written records open for its whole
// life — once those no longer fit the L2s (1,024 resident waves x 124 KB) the records reach HBM as partial lines.  Measured
// (tools/assembly_at_scale.py): twice-refined L_50_R_5, 2.1 GB of records: 2.00 -> 1.18 ms (1.07 -> 1.82 TB/s); L_50_R_5 itself:
// 40.5 -> 36.6 us.  Bit-identical results (tests/test_gpu_parity.py::test_staged_element_stores_give_the_same_bits).  More waves
// per SIMD do not help (the kernel needs all 512 registers: 2 waves spill, direct form 2x slower, staged form unchanged).
constexpr int kStageWords = 64, kStagePitch = 65;   // odd pitch: lane r writes word w at r*65 + w without bank conflicts
template <bool STAGED>
__device__ __forceinline__ void elem_put(double* __restrict__ direct, double* __restrict__ stage, int idx_in_piece, int word, double v) {
  if (STAGED) stage[(threadIdx.x & 63) * kStagePitch + idx_in_piece] = v; else direct[word] = v;
}
// piece [base, base + len) of the records of cells e0 .. e0 + 63 (array `arr`, `stride` doubles per record): out of LDS, coalesced
__device__ __forceinline__ void elem_flush(double* __restrict__ arr, size_t stride, int base, int len, const double* __restrict__ stage, int e0, int nc) {
  __syncthreads();
  const int lane = threadIdx.x & 63;
  if (lane < len) {
    const int nrec = min(64, nc - e0);
#pragma unroll 8
    for (int r = 0; r < nrec; ++r) arr[(size_t)(e0 + r) * stride + base + lane] = stage[r * kStagePitch + lane];
  }
  __syncthreads();
}
#ifndef GMPNP_ELEMENT_WAVES
#define GMPNP_ELEMENT_WAVES 1
#endif
template <int DIM, int NF, bool WANT_J, bool STAGED = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GMPNP_ELEMENT_WAVES, GMPNP_ELEMENT_WAVES))) void k_element(const Ctx c) {
  using L = Lay<DIM, NF>;
  constexpr int NS = L::NS, NN = L::NN;
  static_assert(!STAGED || DIM == 3, "staged record stores: 3D meshes");
  __shared__ double stage[STAGED ? 64 * kStagePitch : 1];
  // Coefficient and quadrature tables go to LDS first: read through the global pointers they would be re-fetched with
  // a vector load (and a full wait) at every use, because the element stores below may alias them.
  __shared__ gmpnp_model_t m;
  __shared__ gmpnp_quadrature_t qd;
  {
    static_assert(sizeof(gmpnp_model_t) % 4 == 0 && sizeof(gmpnp_quadrature_t) % 4 == 0, "word-wise staging");
    const uint32_t* gm = reinterpret_cast<const uint32_t*>(c.model);
    const uint32_t* gq = reinterpret_cast<const uint32_t*>(c.quad);
    uint32_t* lm = reinterpret_cast<uint32_t*>(&m);
    uint32_t* lq = reinterpret_cast<uint32_t*>(&qd);
    for (int w = threadIdx.x; w < (int)(sizeof(gmpnp_model_t) / 4); w += blockDim.x) lm[w] = gm[w];
    for (int w = threadIdx.x; w < (int)(sizeof(gmpnp_quadrature_t) / 4); w += blockDim.x) lq[w] = gq[w];
  }
  __syncthreads();
  const int e0 = blockIdx.x * blockDim.x;
  if (!STAGED && e0 + (int)threadIdx.x >= c.nc) return;
  const int e = min(e0 + (int)threadIdx.x, c.nc - 1);   // STAGED: every lane stays for the copy-out (surplus lanes redo the last cell; never stored)

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
    if constexpr (WANT_J) {  // derivative tables of the monomial: the Jacobian gather reads two numbers per term
      double* dt = c.EJ + (size_t)e * L::EJ_STRIDE + L::O_D + t * 2 * NN * NN;
      const int sp = (t & 1) * 2 * NN * NN;   // STAGED: two terms (2 x 32 words) share a piece
#pragma unroll
      for (int a = 0; a < NN; ++a)
#pragma unroll
        for (int b = 0; b < NN; ++b) {
          elem_put<STAGED>(dt, stage, sp + a * NN + b, a * NN + b, vol * L::KAPPA * (Ys + ys[a] + ys[b] + (a == b ? Ys + 2.0 * ys[a] : 0.0)));            // d/d u_{bj,b}
          elem_put<STAGED>(dt, stage, sp + NN * NN + a * NN + b, NN * NN + a * NN + b, vol * L::KAPPA * (Xs + xs[a] + xs[b] + (a == b ? Xs + 2.0 * xs[a] : 0.0)));  // d/d u_{bk,b}
        }
      if constexpr (STAGED) {
        static_assert(!STAGED || 4 * NN * NN <= kStageWords, "two terms per piece");
        if ((t & 1) || t + 1 == m.n_bilinear)
          elem_flush(c.EJ, L::EJ_STRIDE, L::O_D + (t & ~1) * 2 * NN * NN, ((t & 1) ? 4 : 2) * NN * NN, stage, e0, c.nc);
      }
    }
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
      elem_put<STAGED>(ef, stage, a * NF + i, a * NF + i, f);
      fp += m.qzb[i] * (vol * L::MDEN * (usum[i] + U[a][i]));
    }
    elem_put<STAGED>(ef, stage, a * NF + NS, a * NF + NS, fp);
  }
  if constexpr (STAGED) {
    static_assert(!STAGED || L::EF_STRIDE <= kStageWords, "the residual rows are one piece");
    elem_flush(c.EF, L::EF_STRIDE, 0, L::EF_STRIDE, stage, e0, c.nc);
  }
  // ---- SUPG stabilisation of the PNP model (reference 1D:687-714; 1D meshes only) -----------------------------------
  //   F_stab = - sum_i rho_i z_i [ (u_i - u_i^n)/(dt L_D) + z_i grad(w_i).grad(p) + R_i ] grad(p).grad(v_i) dx
  // rho_i: nodal (P1), w_i = u_i except the reference's OH term, which takes grad(u_H) (c.supg_w); R_i the
  // production rate (the tables hold -R_i).  Degree <= 3 on a P1 element: closed form.  The element matrix of these
  // terms is dense in (species, potential) and is stored whole (196 doubles) for the Jacobian gather.
  if constexpr (DIM == 1) {
    if (c.supg_rho) {
      double* js = nullptr;
      if constexpr (WANT_J) {
        js = c.EJ + (size_t)e * L::EJ_STRIDE + L::O_S;
        for (int q = 0; q < NN * NF * NN * NF; ++q) js[q] = 0.0;
      }
      for (int i = 0; i < NS; ++i) {
        const double zi = m.z[i];
        if (zi == 0.0) continue;
        const int wi = c.supg_w[i];
        double rho[NN], du[NN], uw[NN];
#pragma unroll
        for (int a = 0; a < NN; ++a) {
          rho[a] = c.supg_rho[(size_t)nd[a] * NS + i];
          du[a] = U[a][i] - c.un[(size_t)nd[a] * NF + i];
          double v = 0.0;
#pragma unroll
          for (int j = 0; j < NS; ++j) v = (j == wi) ? U[a][j] : v;
          uw[a] = v;
        }
        double rsum = 0.0, gradw = 0.0;
#pragma unroll
        for (int a = 0; a < NN; ++a) { rsum += rho[a]; gradw += uw[a] * g[a][0]; }
        const double rbar = rsum * (1.0 / NN);
        double rM[NN];
#pragma unroll
        for (int b = 0; b < NN; ++b) rM[b] = vol * L::MDEN * (rsum + rho[b]);
        double S = zi * (gradw * gradp[0]) * vol * rbar - m.rc0[i] * vol * rbar;
#pragma unroll
        for (int b = 0; b < NN; ++b) S += m.inv_dt * rM[b] * du[b];
        for (int j = 0; j < NS; ++j) {
          const double c1 = m.rc1[i][j];
          if (c1 != 0.0)
#pragma unroll
            for (int b = 0; b < NN; ++b) S -= c1 * rM[b] * U[b][j];
        }
        // bilinear terms: sum_abc rho_a x_b y_c T_abc |K|, T_abc = kappa (6 | 2 | 1 for three | two | no equal indices)
        double dSx[GMPNP_MAX_BILINEAR][NN], dSy[GMPNP_MAX_BILINEAR][NN];
        for (int t = 0; t < m.n_bilinear; ++t) {
          const double c2 = m.rc2[i][t];
          const int bj = m.bil_j[t], bk = m.bil_k[t];
          double xs[NN], ys[NN];
#pragma unroll
          for (int a = 0; a < NN; ++a) {
            double xv = 0.0, yv = 0.0;
#pragma unroll
            for (int j = 0; j < NS; ++j) { xv = (j == bj) ? U[a][j] : xv; yv = (j == bk) ? U[a][j] : yv; }
            xs[a] = xv; ys[a] = yv;
          }
          double tot = 0.0;
#pragma unroll
          for (int b = 0; b < NN; ++b) { dSx[t][b] = 0.0; dSy[t][b] = 0.0; }
#pragma unroll
          for (int a = 0; a < NN; ++a)
#pragma unroll
            for (int b = 0; b < NN; ++b)
#pragma unroll
              for (int cc = 0; cc < NN; ++cc) {
                const double T = vol * L::KAPPA * ((a == b && b == cc) ? 6.0 : ((a == b || b == cc || a == cc) ? 2.0 : 1.0));
                tot += rho[a] * xs[b] * ys[cc] * T;
                dSx[t][b] += rho[a] * ys[cc] * T;   // d/d x_b
                dSy[t][cc] += rho[a] * xs[b] * T;   // d/d y_c
              }
          S -= c2 * tot;
        }
#pragma unroll
        for (int a = 0; a < NN; ++a) ef[a * NF + i] += -zi * gp[a] * S;
        if constexpr (WANT_J) {
#pragma unroll
          for (int b = 0; b < NN; ++b) {
            double dS[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) dS[j] = 0.0;
            for (int j = 0; j < NS; ++j) {
              double v = -m.rc1[i][j] * rM[b];
              if (j == i) v += m.inv_dt * rM[b];
              if (j == wi) v += zi * gp[b] * vol * rbar;
              for (int t = 0; t < m.n_bilinear; ++t) {
                const double c2 = m.rc2[i][t];
                if (j == m.bil_j[t]) v -= c2 * dSx[t][b];
                if (j == m.bil_k[t]) v -= c2 * dSy[t][b];
              }
              dS[j] = v;
            }
            const double dSp = zi * (gradw * g[b][0]) * vol * rbar;
#pragma unroll
            for (int a = 0; a < NN; ++a) {
              double* row = js + ((size_t)(a * NF + i) * NN + b) * NF;
              for (int j = 0; j < NS; ++j) row[j] = -zi * gp[a] * dS[j];
              row[NS] = -zi * (gg[a][b] * S + gp[a] * dSp);
            }
          }
        }
      }
    }
  }
  if constexpr (WANT_J) {
    double* ej = c.EJ + (size_t)e * L::EJ_STRIDE;
    // two pieces: [0, O_B) = volume, gradients, means, eps, int u beta; [O_B, O_D) = int beta phi, int u beta^2 phi
    elem_put<STAGED>(ej, stage, L::O_VOL, L::O_VOL, vol);
#pragma unroll
    for (int a = 0; a < NN; ++a) {
#pragma unroll
      for (int b = 0; b < NN; ++b) elem_put<STAGED>(ej, stage, L::O_GG + a * NN + b, L::O_GG + a * NN + b, gg[a][b]);
      elem_put<STAGED>(ej, stage, L::O_GP + a, L::O_GP + a, gp[a]);
      elem_put<STAGED>(ej, stage, L::O_GG_A + a, L::O_GG_A + a, Gg[a]);
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      elem_put<STAGED>(ej, stage, L::O_UBAR + j, L::O_UBAR + j, ubar[j]);
      elem_put<STAGED>(ej, stage, L::O_IJ + j, L::O_IJ + j, Ij[j]);
    }
    elem_put<STAGED>(ej, stage, L::O_EPS, L::O_EPS, epsbar);
    if constexpr (STAGED) {
      static_assert(!STAGED || (L::O_B <= kStageWords && L::O_D - L::O_B <= kStageWords), "the record head is two pieces");
      elem_flush(c.EJ, L::EJ_STRIDE, 0, L::O_B, stage, e0, c.nc);
    }
#pragma unroll
    for (int a = 0; a < NN; ++a) elem_put<STAGED>(ej, stage, a, L::O_B + a, Bq[a]);
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
      for (int b = 0; b < NN; ++b) elem_put<STAGED>(ej, stage, (L::O_C - L::O_B) + j * NN + b, L::O_C + j * NN + b, Cq[j][b]);
    if constexpr (STAGED) elem_flush(c.EJ, L::EJ_STRIDE, L::O_B, L::O_D - L::O_B, stage, e0, c.nc);
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
    const int Ig = r / NF;
    if (Ig < c.own_node0 || Ig >= c.own_node1) {
      val = 0.0;   // ghost row of a partitioned handle: the owner's rank holds it
    } else if (c.bcflag[r]) {
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
    c.kr[r] = val; c.kb[r] = val;   // the next linear solve's right-hand side, where BiCGStab expects it (no copy launch)
  }
  double v[1] = {val * val};
  block_sum<1>(v, lds);
  if (threadIdx.x == 0) {
    c.part_f[blockIdx.x] = v[0];
    if (blockIdx.x == 0) c.status_host[0] = __hip_atomic_load(c.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // flags of the kernels before this one
  }
}

// ---------------------------------------------------------------------------------------------
// Jacobian gather: one wave per (slice, kpos); lane = (block row in slice, scalar row i) computes the
// NF entries of its row of block (I, cols[kpos]) by summing the element contributions in element
// order and stores them at stride 64 doubles (each store instruction writes 63 contiguous doubles).
// ---------------------------------------------------------------------------------------------
// One wave per (slice, block position), as many waves in flight as there are blocks (the kernel is latency bound: a
// workgroup per slice looping over its positions moved 3x fewer bytes and took 1.2-1.8x longer).  The work list runs slice
// by slice, and blockIdx is remapped so that a contiguous run of the list lands on ONE XCD (workgroups are dealt to the 8
// XCDs round robin): the element records around a slice's nodes are then shared through that XCD's L2 by the ~15 waves
// of the slice and by the neighbouring slices, instead of being fetched by every XCD.
constexpr int kXcds = 8;
// wave index into the work list: equal runs, a multiple of 8 of them (gmpnp_topology.cpp); XCD x takes runs x, x + 8, ...
__device__ __forceinline__ int xcd_run_wave(const Ctx& c) {
  const int x = (int)blockIdx.x % kXcds, j = (int)blockIdx.x / kXcds;          // XCD, position in that XCD's queue
  const int run = x + kXcds * (j / c.wl_run_blocks), off = j - (j / c.wl_run_blocks) * c.wl_run_blocks;
  return ((run * c.wl_run_blocks + off) * kVecBlock + (int)threadIdx.x) >> 6;
}
template <int DIM, int NF>
__global__ __launch_bounds__(kVecBlock) void k_jac_gather(const Ctx c) {
  using L = Lay<DIM, NF>;
  constexpr int NS = L::NS, NN = L::NN, G = 2, MB = GMPNP_MAX_BILINEAR;
  const int wave = xcd_run_wave(c), lane = threadIdx.x & 63;
  if (wave >= c.n_work) return;
  const int s = c.wl_slice[wave], kpos = c.wl_kpos[wave];
  if (s < 0) return;   // padding of a run
  const int Iloc = lane / NF, i = lane - Iloc * NF;
  if (Iloc >= c.slice_nn[s]) return;
  const int I = c.slice_node0[s] + Iloc;
  const int k = c.sell_blk[(size_t)(c.slice_colbase[s] + kpos) * kSlicePad + Iloc];
  if (k < 0) return;  // padding stays zero (set at create)
  const int J = c.cols[k];
  const int qb = c.cptr[k], qend = c.cptr[k + 1];
  const int bc = c.bcflag[I * NF + i];
  double* out = c.vals + c.slice_off[s] + (size_t)kpos * NF * kWave + lane;
  const gmpnp_model_t& m = *c.model;
  const bool isp = (i == NS);
  const int is = isp ? 0 : i;
  const double zi = m.z[is], inv_dt = m.inv_dt;
  double rc1i[NS], c2t[MB];
#pragma unroll
  for (int j = 0; j < NS; ++j) rc1i[j] = m.rc1[is][j];
#pragma unroll
  for (int t = 0; t < MB; ++t) { const double v = m.rc2[is][t]; c2t[t] = (t < m.n_bilinear && !isp) ? v : 0.0; }
  const int tmax = max(m.n_bilinear, 1) - 1;
  const int qe = bc ? qb : qend;  // Dirichlet rows take no contributions
  double acc[NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) acc[j] = 0.0;

  // G contributions per trip; every value of a trip is requested before the first one is used (unconditional loads on
  // clamped indices, masked by w = 0/1), and the contribution codes of the NEXT trip are requested with them: one
  // memory round trip per trip instead of one per table.
  int pk[G];
#pragma unroll
  for (int u = 0; u < G; ++u) pk[u] = c.contrib[max(min(qb + u, qe - 1), 0)];
  for (int q0 = qb; q0 < qe; q0 += G) {
    double vol[G], ggab[G], gpa[G], gga[G], cq[G], ij[G], bq[G], ub[G], ep[G], dj[G][MB], dk[G][MB];
    int pa[G], pb[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int e = pk[u] >> 4, a = (pk[u] >> 2) & 3, b = pk[u] & 3;
      const double* ej = c.EJ + (size_t)e * L::EJ_STRIDE;
      pa[u] = a; pb[u] = b;
      vol[u] = ej[L::O_VOL]; ggab[u] = ej[L::O_GG + a * NN + b]; gpa[u] = ej[L::O_GP + a]; gga[u] = ej[L::O_GG_A + a];
      cq[u] = ej[L::O_C + is * NN + b]; ij[u] = ej[L::O_IJ + is]; bq[u] = ej[L::O_B + b]; ub[u] = ej[L::O_UBAR + is];
      ep[u] = ej[L::O_EPS];
#pragma unroll
      for (int t = 0; t < MB; ++t) {
        const double* dt = ej + L::O_D + min(t, tmax) * 2 * NN * NN + a * NN + b;
        dj[u][t] = dt[0]; dk[u][t] = dt[NN * NN];
      }
    }
    int pkn[G];
#pragma unroll
    for (int u = 0; u < G; ++u) pkn[u] = c.contrib[max(min(q0 + G + u, qe - 1), 0)];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const double w = (q0 + u < qe) ? 1.0 : 0.0;
      const double Mab = vol[u] * L::MDEN * (pa[u] == pb[u] ? 2.0 : 1.0), Kab = vol[u] * ggab[u];
      if (!isp) {
        const double ster = gga[u] * cq[u] + ij[u] * ggab[u];
        const double dg = inv_dt * Mab + Kab + zi * vol[u] * (1.0 / NN) * gpa[u] + gga[u] * bq[u];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          double term = m.a[j] * ster + rc1i[j] * Mab + (j == is ? dg : 0.0);
#pragma unroll
          for (int t = 0; t < MB; ++t)  // c2t = 0 beyond n_bilinear
            term += (j == m.bil_j[t] ? c2t[t] * dj[u][t] : 0.0) + (j == m.bil_k[t] ? c2t[t] * dk[u][t] : 0.0);
          acc[j] += w * term;
        }
        acc[NS] += w * (zi * ub[u] * Kab);
      } else {
        const double kpa = vol[u] * gpa[u] * (1.0 / NN);
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[j] += w * (-m.epsc[j] * kpa + m.qzb[j] * Mab);
        acc[NS] += w * (-ep[u] * Kab);
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) pk[u] = pkn[u];
  }
  if constexpr (DIM == 1) {
    if (c.supg_rho) {  // dense SUPG element matrices (PNP + stabilisation), added after the regular terms in element order
      for (int q = qb; q < qe; ++q) {
        const int pk = c.contrib[q];
        const int e = pk >> 4, a = (pk >> 2) & 3, b = pk & 3;
        const double* row = c.EJ + (size_t)e * L::EJ_STRIDE + L::O_S + ((size_t)(a * NF + i) * NN + b) * NF;
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[j] += row[j];
      }
    }
  }
  if (bc) {  // [3P] DirichletBC.apply(A): identity row
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[j] = (J == I && j == i) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int j = 0; j < NF; ++j) out[j * kWave] = acc[j];
}

__global__ void k_robin_add(const Ctx c) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= c.n_robin) return;
  if (c.bcflag[c.rob_row[t]]) return;
  c.vals[c.rob_addr[t]] += c.rob_val[t];
}

// ---------------------------------------------------------------------------------------------
// Node-block Jacobi: invert the NF x NF diagonal blocks (Gauss-Jordan, partial pivoting, LDS-resident).
// ---------------------------------------------------------------------------------------------
// Dinv = inverse of the diagonal node blocks.  16 lanes per node: lane r (< NF) holds row r of [A | I] in registers;
// Gauss-Jordan with partial pivoting, pivot search and row broadcasts by shuffles inside the 16-lane group.
template <int NF>
__global__ __launch_bounds__(64) void k_block_inverse(const Ctx c) {
  const int t = threadIdx.x, r = t & 15, I_raw = blockIdx.x * 4 + (t >> 4);
  const bool on = I_raw < c.nv;
  const int I = on ? I_raw : 0, rc = r < NF ? r : 0;
  const int s = c.node_slice[I], Iloc = I - c.slice_node0[s];
  const double* base = c.vals + c.slice_off[s] + Iloc * NF + rc;  // the diagonal block is SELL position 0
  double row[2 * NF];
#pragma unroll
  for (int j = 0; j < NF; ++j) { const double v = base[(size_t)j * kWave]; row[j] = (r < NF) ? v : 0.0; row[NF + j] = (j == r) ? 1.0 : 0.0; }
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
    // rows k and idx swap places (lane k takes the pivot row, lane idx the old row k), then scale / eliminate
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
  if (on && bad) { atomicOr(c.status, 2); }
  if (on && r < NF) {
    double* o = c.Dinv + ((size_t)I * NF + r) * NF;
#pragma unroll
    for (int j = 0; j < NF; ++j) o[j] = row[NF + j];
  }
}

template <int NF>
__global__ __launch_bounds__(64) void k_coarse_rows(const Ctx c) {
  constexpr int S = kWave / NF;
  const int s = blockIdx.x, lane = threadIdx.x;
  const int Iloc = lane / NF, i = lane - Iloc * NF;
  if (Iloc >= c.slice_nn[s]) return;
  const int I = c.slice_node0[s] + Iloc;
  double acc[kMaxRowAggs][NF];
#pragma unroll
  for (int q = 0; q < kMaxRowAggs; ++q)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[q][j] = 0.0;
  const int cb = c.slice_colbase[s], mx = c.slice_colbase[s + 1] - cb;
  const double* base = c.vals_s + c.slice_off[s] + lane;  // the coarse operator is built from the column-scaled matrix
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
  const int nch = c.coarse_chunks, g = blockIdx.x / nch, ch = blockIdx.x - g * nch, n = c.ncoarse;
  const int a0 = c.agg_start[g], len = c.agg_start[g + 1] - a0;
  const int I0 = a0 + (int)((int64_t)len * ch / nch), I1 = a0 + (int)((int64_t)len * (ch + 1) / nch);
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
  for (int ch = 0; ch < c.coarse_chunks; ++ch) s += c.AcPart[(size_t)ch * n2 + q];
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
  for (int q = t; q < n * n; q += nt) c.Aci[q] = A[q];
}

// =================================================================================================
// Fused BiCGStab.  System: A Dinv (I + P Aci P^T) y = b, x = Dinv (I + P Aci P^T) y, with
//   As  = A Dinv          the column-scaled Jacobian (k_scale_columns, once per Newton iteration)
//   Aci = (P^T As P)^-1   the inverse Galerkin operator on the slab aggregates.
// One iteration is TWO launches.  Workgroup = one tile (kSlicesPerTile slices of one aggregate, 8 waves per slice).
//   A(k): omega, beta from B(k-1)'s partials; own rows: y += alpha p + omega s, r = s - omega t, p = r + beta (p - omega v);
//         the SpMV gathers p at the column nodes ON THE FLY from (s, t, p_old, v_old), adds the prolonged coarse
//         correction and produces v = As (p + P Aci P^T p); epilogue: (rhat,v), ||r||^2, partial restriction of v.
//   B(k): convergence test, alpha; own rows s = r - alpha v; SpMV gathers s from (r, v): t = As (s + P Aci P^T s);
//         epilogue: (t,s) (t,t) (rhat,s) (rhat,t), partial restriction of t.
// The restrictions P^T p and P^T s are never formed from the fine vectors: they follow the same recurrences on the
// coarse level (P^T r, P^T p kept in crc/cpc; P^T v, P^T t come from the kernels' epilogue partials), so the coarse
// solve needs no reduction of its own and every workgroup evaluates just the coarse rows its columns prolong from.
// All reductions are fixed-order sums of per-tile partials: results are bitwise reproducible.
// =================================================================================================
constexpr int kKrylovWaves = GMPNP_KRYLOV_WAVES;                          // waves per slice
constexpr int kKrylovThreads = kSlicesPerTile * kKrylovWaves * kWave;    // 512
template <int NF> constexpr int stage_count() { return (kTileCols * NF + kKrylovThreads - 1) / kKrylovThreads; }

constexpr int kCoarseThreads = 512;  // coarse kernels: one wave per coarse row, 8 rows per workgroup

// ---- prologue pieces.  Each has a load() that only ISSUES global loads into registers (called first thing in the
// kernel, so every request of the launch is in flight together) and a later reduce step on LDS.
// Every load is UNCONDITIONAL on a clamped (always valid) index and masked afterwards: a predicated load becomes a
// branch, the compiler sinks the first use into it and waits there, and the prologue degenerates into one memory
// round trip per load.

// K partial arrays of n entries each (part + q*stride) -> totals for every thread.
template <int K>
struct PartialSums {
  static constexpr int U = 2;  // entries per thread requested up front: n <= U * kCoarseThreads, else the tail loop
  double r[K][U];
  __device__ inline void load(const double* __restrict__ part, int n, int stride) {  // straight-line: requests only
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ic = min((int)threadIdx.x + u * kCoarseThreads, n - 1);
#pragma unroll
      for (int q = 0; q < K; ++q) r[q][u] = part[(size_t)q * stride + ic];
    }
  }
  // lds: [ (kCoarseThreads/64) * K ].  reduce_partial: per-wave sums into LDS; the CALLER synchronises; reduce_final: totals.
  __device__ inline void reduce_partial(double* lds, const double* __restrict__ part, int n, int stride) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double v[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
      double sacc = 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u) sacc += ((int)threadIdx.x + u * kCoarseThreads < n) ? r[q][u] : 0.0;
      v[q] = sacc;
    }
    for (int i0 = threadIdx.x + U * kCoarseThreads; i0 < n; i0 += 4 * kCoarseThreads) {  // large meshes only
      double wv_[K][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < K; ++q) wv_[q][u] = part[(size_t)q * stride + min(i0 + u * kCoarseThreads, n - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] += (i0 + u * kCoarseThreads < n) ? wv_[q][u] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] = wave_sum(v[q]);
    if (lane == 0)
#pragma unroll
      for (int q = 0; q < K; ++q) lds[q * NWV + w] = v[q];
  }
  // Totals for every thread: lane 8q + w reads wave w's sum of quantity q (ONE LDS read per lane), three DPP row shifts put the
  // total of quantity q into lane 8q + 7, a readlane makes it uniform.  (Reading all 8 x K per-wave sums into every thread
  // costs 16 K registers at the point where the coarse workgroups hold their Aci columns: the fused A launch spilled there.)
  static constexpr int NWV = kCoarseThreads / 64;
  static_assert(NWV == 8 && 8 * K <= 64, "one 8-lane group per quantity");
  static __device__ inline void reduce_final(const double* lds, double (&out)[K]) {
    const int lane = threadIdx.x & 63;
    double v = lds[min(lane, 8 * K - 1)];
    v = group8_sum_to_last(v);
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(v), 8 * q + 7), hi = __builtin_amdgcn_readlane(__double2hiint(v), 8 * q + 7);
      out[q] = __hiloint2double(hi, lo);
    }
  }
};

// Layout of the per-tile restriction partials: part[aggregate][slot][field] — the tile_slots x NF values of ONE aggregate
// contiguous (38 cache lines for 66 slots), because its coarse workgroup is the one that reads them and does so on every
// tile's critical path.  (part[slot][coarse dof] put an aggregate's 72-byte pieces 576 bytes apart: 66-132 lines per array.)
__device__ __forceinline__ size_t cpart_index(const Ctx& c, int slot, int agg, int f, int nf) {
  return ((size_t)agg * c.tile_slots + slot) * nf + f;
}
// Restriction partials of ONE aggregate g: 8 lanes per (array q, field f) sum the slots of coarse dof
// g*NF+f; the workgroup of aggregate g needs nothing else of the partial arrays.
template <int K, int NF>
struct AggSlotSums {
  static constexpr int L = 8, U = 9;  // lanes per dof, slots per lane requested up front (72 slots, then the tail loop)
  double r[U];
  const double* base;
  __device__ inline void load(const double* const (&part)[K], int n, int slots, int g) {  // straight-line: requests only
    const int t = threadIdx.x, combo = min(t / L, K * NF - 1), l = t & (L - 1);
    const int q = combo / NF, f = combo - q * NF;
    base = part[0];
#pragma unroll
    for (int z = 1; z < K; ++z) base = (q == z) ? part[z] : base;
    base += (size_t)g * slots * NF + f;
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = base[(size_t)min(l + L * u, slots - 1) * NF];
  }
  // lds[K*NF]; the caller synchronises
  __device__ inline void to_lds(double* lds, int n, int slots) {
    const int t = threadIdx.x, combo = t / L, l = t & (L - 1);
    double v = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) v += (l + L * u < slots) ? r[u] : 0.0;
    for (int s0 = l + L * U; s0 < slots; s0 += 4 * L) {  // large meshes only
      double w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w[u] = base[(size_t)min(s0 + u * L, slots - 1) * NF];
#pragma unroll
      for (int u = 0; u < 4; ++u) v += (s0 + u * L < slots) ? w[u] : 0.0;
    }
    static_assert(L == 8, "group8_sum_to_last");
    v = group8_sum_to_last(v);
    if (l == L - 1 && combo < K * NF) lds[combo] = v;
  }
};

// yc of the coarse dofs a tile prolongs from = sum over the column-block partials yc[g][dof] the coarse workgroups
// wrote: 8 lanes per dof (aggregate g = lane, lane + 8; nagg <= 16), kTileAggs*NF <= 64 dofs in one pass (padding
// aggregates repeat aggregate 0).
template <int NF>
struct TileCoarse {
  static constexpr int ROWS = kTileAggs * NF;
  static_assert(ROWS * 8 <= kKrylovThreads, "one pass: 8 lanes per coarse dof of the tile");
  double a0, a1;
  int d, nrows;
  __device__ inline void load_index(const Ctx& c, int tile) {  // straight-line: requests only
    const int t = threadIdx.x, rc = min(t >> 3, ROWS - 1);
    const int ag = c.tile_aggs[tile * kTileAggs + rc / NF];
    d = ag * NF + (rc - (rc / NF) * NF);
    nrows = c.tile_nagg[tile] * NF;   // coarse dofs the tile really prolongs from (the other slots repeat aggregate 0)
  }
  template <bool COHERENT>
  __device__ inline void load_values(const Ctx& c) {
    const int n = c.ncoarse, l = threadIdx.x & 7;
    const double* p0 = c.yc + (size_t)min(l, c.nagg - 1) * n + d;
    const double* p1 = c.yc + (size_t)min(l + 8, c.nagg - 1) * n + d;
    if (COHERENT) {  // written by coarse workgroups of the same launch: agent-scope loads, no cached copy
      // Behind the hand-over every tile asks for these lines in the same microsecond: only the requests that carry something —
      // the tile's real aggregates (2-3 of the kTileAggs slots), and the second column-block half only with more than 8 blocks.
      a0 = 0.0; a1 = 0.0;
      if ((int)(threadIdx.x >> 3) < nrows) {
        a0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c.nagg > 8) a1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else { a0 = *p0; a1 = *p1; }
  }
  // ycl[kMaxCoarse], indexed by the global coarse dof; the caller synchronises
  __device__ inline void to_lds(const Ctx& c, double* ycl) {
    const int t = threadIdx.x, l = t & 7;
    double a = ((l < c.nagg) ? a0 : 0.0) + ((l + 8 < c.nagg) ? a1 : 0.0);
    a = group8_sum_to_last(a);
    if (l == 7 && (t >> 3) < nrows) ycl[d] = a;   // (the padding slots repeat aggregate 0: they must not overwrite its sums)
  }
};

// Rows of Aci this tile needs (tile_nagg*NF <= kTileAggs*NF rows), 16 lanes per row, two row passes per thread.
template <int NF>
struct CoarseRows {
  static constexpr int U = 9;  // ceil(kMaxCoarse / 16)
  static constexpr int PASS = (kTileAggs * NF + kKrylovThreads / 16 - 1) / (kKrylovThreads / 16);
  double a[PASS][U];
  __device__ inline void load(const Ctx& c, int tile) {
    const int n = c.ncoarse, t = threadIdx.x, nrows = c.tile_nagg[tile] * NF;
#pragma unroll
    for (int ps = 0; ps < PASS; ++ps) {
      const int r = (t >> 4) + ps * (kKrylovThreads >> 4);
      const bool on = r < nrows;
      const int slot = on ? r / NF : 0, f = r - slot * NF;
      const double* arow = c.Aci + (size_t)(c.tile_aggs[tile * kTileAggs + slot] * NF + (on ? f : 0)) * n;
#pragma unroll
      for (int u = 0; u < U; ++u) { const int cc = (t & 15) + 16 * u; a[ps][u] = (on && cc < n) ? arow[cc] : 0.0; }
    }
  }
  __device__ inline void apply(const Ctx& c, int tile, const double* pcs, double* ycl) const {
    const int n = c.ncoarse, t = threadIdx.x, nrows = c.tile_nagg[tile] * NF;
#pragma unroll
    for (int ps = 0; ps < PASS; ++ps) {
      const int r = (t >> 4) + ps * (kKrylovThreads >> 4);
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u) { const int cc = (t & 15) + 16 * u; if (cc < n) acc += a[ps][u] * pcs[cc]; }
      acc += __shfl_xor(acc, 1, 16); acc += __shfl_xor(acc, 2, 16); acc += __shfl_xor(acc, 4, 16); acc += __shfl_xor(acc, 8, 16);
      if ((t & 15) == 0 && r < nrows) ycl[r] = acc;
    }
  }
};

// Load base[byte_off / 8] with a 32-bit byte offset from a wave-uniform base: the scalar-base addressing form, one VGPR per
// address instead of a 64-bit pair (the tile prologues hold ~50 load destinations at once and have 80 registers).
__device__ __forceinline__ double ld_off(const double* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
}

// Matrix part shared by the tile kernels: the wave's first PRE blocks are requested up front, x comes from LDS.
// The preload is unconditional (vals / sell_lcol carry kRowPad block positions of zero padding behind the last
// slice, positions past a short slice's end just read into the next slice) and masked afterwards.
template <int NF, int PRE_ = GMPNP_ROW_PRELOAD>
struct TileRows {
  static constexpr int PRE = PRE_;
  double av[PRE][NF];
  int lc[PRE];
  int Iloc, i, cb, mx, w, row;
  bool active;
  const double* base;

  __device__ inline void load(const Ctx& c, const double* __restrict__ vals, const TileRec& rec) {
    const int lane = threadIdx.x & 63;
    w = threadIdx.x >> 6;
    Iloc = lane / NF; i = lane - Iloc * NF;
    active = Iloc < rec.nn;
    row = active ? (rec.node0 + Iloc) * NF + i : 0;
    cb = rec.colbase; mx = active ? rec.mx : 0; base = vals + rec.slice_off + lane;
    const int il = min(Iloc, kSlicePad - 1);
    // Positions past the slice's last block are clamped onto it: the request repeats a line another wave of this tile asks
    // for anyway instead of pulling a line of the NEXT slice through the fabric (every launch re-reads its bytes from the
    // memory side — the XCDs' L2s keep nothing across a kernel boundary — and the tile kernels are bound by those bytes:
    // the unclamped two-position preload of the B half fetched 6.8 MB more than the matrix holds).
    const int last = GMPNP_CLAMP_PRELOAD ? rec.mx - 1 : (1 << 30);
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
      const int kp = min(w + u * kKrylovWaves, last);
      lc[u] = c.sell_lcol[(size_t)(cb + kp) * kSlicePad + il];
    }
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
      const int kp = min(w + u * kKrylovWaves, last);
#pragma unroll
      for (int j = 0; j < NF; ++j) av[u][j] = base[(size_t)(kp * NF + j) * kWave];
    }
  }

  __device__ inline double dot(const Ctx& c, const double* xs) const {
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
      const bool on = w + u * kKrylovWaves < mx;
      const double* xv = xs + (on ? lc[u] : 0) * NF;
#pragma unroll
      for (int j = 0; j < NF; ++j) acc += (on ? av[u][j] : 0.0) * xv[j];
    }
    if (active) {
      for (int kp = w + PRE * kKrylovWaves; kp < mx; kp += kKrylovWaves) {  // rows with more than PRE*8 blocks
        const double* xv = xs + c.sell_lcol[(size_t)(cb + kp) * kSlicePad + Iloc] * NF;
        const double* ap = base + (size_t)kp * NF * kWave;
#pragma unroll
        for (int j = 0; j < NF; ++j) acc += ap[(size_t)j * kWave] * xv[j];
      }
    }
    return acc;
  }
};

constexpr int kStagePre = 2;  // staged x entries per thread requested up front (tiles with more columns: tail loop)

#ifdef GMPNP_TIMING  // development builds only: phase timestamps (100 MHz clock) of three tiles behind the yc blocks
#define GMPNP_STAMP(i) do { if (threadIdx.x == 0) { const int tq_ = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 ? 1 : (blockIdx.x == gridDim.x - 1 ? 2 : -1)); \
    if (tq_ >= 0) c.yc[kMaxCoarse * 16 + tq_ * 8 + (i)] = (double)wall_clock64(); } } while (0)
#else
#define GMPNP_STAMP(i) do { } while (0)
#endif
#ifdef GMPNP_XTIMING  // development builds only: phase timestamps of the exchange-prologue launches of iteration 5 (tools/xch_phases.py)
#define GMPNP_XSTAMP(k_, i) do { if (threadIdx.x == 0 && (k_) == 5) c.yc[kMaxCoarse * 16 + 32 + (i)] = (double)wall_clock64(); } while (0)
#else
#define GMPNP_XSTAMP(k_, i) do { } while (0)
#endif
// Early exit of a finished solve.  The requested values get a (never executed) use on the exit path: without it the
// compiler sinks every request below this branch, i.e. behind the scalar round trip that fetches the flag.
#define GMPNP_EXIT_IF_DONE(flag, keep_expr) do { if (flag) { if (c.ndof < 0) c.yc[0] = (keep_expr); return; } } while (0)
// ---- in-launch hand-over from the coarse workgroups to the tile workgroups (fused launch form) ---------------------
#ifndef GMPNP_FLAG_COPIES
#define GMPNP_FLAG_COPIES 8
#endif
constexpr int kFlagCopies = GMPNP_FLAG_COPIES;   // copies of the flag row (power of two, <= 64: 16 words apart, Ctx::ticket holds 16 * 66)
// Hand-over flags: coarse workgroup g raises flag g (one cache line holds all of them) to the sequence number of the
// launch; lane g of a tile workgroup's first wave polls flag g and the wave leaves the loop on a unanimous vote.  No
// counter, no read-modify-write, no second hop: the consumers see a coarse workgroup's flag one store after its payload.
template <bool FUSED>
__device__ __forceinline__ void publish_ticket(const Ctx& c, unsigned seq, int g) {
  if (FUSED) {
    // No fences: the payload (yc, scalars) went out as agent-scope write-through stores (store_coherent); every thread
    // drains its own stores, then one thread raises the flag.  A release fence would write back the whole L2 of the
    // XCD and the consumers' acquire would invalidate theirs 66 times per launch (measured: +7 us per launch).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // eight copies of the flag row (one 64-byte line each): a tile polls the copy of its XCD's number, so that 66 instead of 529
    // workgroups hammer one line
    if (threadIdx.x < kFlagCopies) __hip_atomic_store(c.ticket + threadIdx.x * 16 + g, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// progress / verdict for the host's spin loop (HostPoll): write-through to system memory, no fence
__device__ __forceinline__ void poll_store(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void poll_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// end of the solve: rr and iters first, the exit code behind a system-scope release (once per solve)
__device__ __forceinline__ void poll_finish(const Ctx& c, double rr, int iters, int done) {
  poll_store(&c.poll->rr, rr); poll_store(&c.poll->iters, iters);
  __hip_atomic_store(&c.poll->done, done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool wait_ticket(const Ctx& c, unsigned seq, unsigned long long budget = 200000000ull) {
  __shared__ int ticket_ok;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const bool mine = lane < c.nagg;
    int ok = 1;
    const unsigned long long t0 = wall_clock64();
    while (true) {
      const unsigned f = mine ? __hip_atomic_load(c.ticket + (blockIdx.x & (kFlagCopies - 1)) * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seq;
      if (__all(f >= seq)) break;
      // 2 s at 100 MHz.  The flags cannot fail to arrive (coarse workgroups are dispatched first and wait for nobody); the
      // budget only has to outlast a time slice taken by another process sharing the GPU.  Ends the solve (later
      // launches exit at once).
      if (wall_clock64() - t0 > budget) { ok = 0; if (lane == 0) { atomicOr(c.status, 8); c.scal->done = 3; poll_finish(c, c.scal->rr, c.scal->iters, 3); } break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) ticket_ok = ok;
  }
  __syncthreads();
  return ticket_ok != 0;
}
// scalar written by another workgroup of the SAME launch: a vector load at agent scope (a scalar load could hit a stale
// line of the scalar cache, which earlier reads of the same struct pulled in)
__device__ __forceinline__ double load_coherent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool FUSED>
__device__ __forceinline__ void store_coherent(double* p, double v) {
  if (FUSED) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool FUSED>
__device__ __forceinline__ void store_coherent(int32_t* p, int32_t v) {
  if (FUSED) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ double load_coherent(const int32_t* p) {
  return (double)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exchange-as-prologue launches (XCH; gmpnp_dist_kernels.h).  What the ranks send each other there travels as FLAGGED WORDS: a
// double is two 8-byte words, each = 32 bits of it | the exchange's sequence number << 32, stored with one 8-byte system-scope store
// into the receiver's mailbox (uncached memory).  A word is valid when its upper half equals the sequence number the reader expects:
// no flag behind the data, no store drain, no fence, no counter — data and validity are ONE atomic object (the low-latency protocol
// of the collective libraries).  A reader polls exactly the words it needs.  Budget: 12 s at 100 MHz (a rank that is gone); on
// expiry the solve ends with status bit 8, like a hand-over that never comes.
constexpr unsigned long long kXchBudget = 1200000000ull;
__device__ __forceinline__ void ll_store(unsigned long long* dst, double v, uint32_t seq) {
  const unsigned long long tag = (unsigned long long)seq << 32;
  __hip_atomic_store(dst, tag | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(dst + 1, tag | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// N doubles at once: all 2N words are requested together and re-requested together until every one carries its sequence number
// (one memory round trip when everything has arrived, whatever N)
template <int N>
__device__ __forceinline__ void ll_wait_n(const Ctx& c, const unsigned long long* const (&src)[N], const uint32_t (&seq)[N], double (&out)[N]) {
  unsigned long long w[N][2];
  const unsigned long long t0 = wall_clock64();
  while (true) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      w[i][0] = __hip_atomic_load(src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      w[i][1] = __hip_atomic_load(src[i] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) ok = ok && (uint32_t)(w[i][0] >> 32) == seq[i] && (uint32_t)(w[i][1] >> 32) == seq[i];
    if (ok) break;
    if (wall_clock64() - t0 > kXchBudget) { atomicOr(c.status, 8); c.scal->done = 3; break; }
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = __hiloint2double((int)(uint32_t)w[i][1], (int)(uint32_t)w[i][0]);
}
// where rank q's contribution to entry i of the sums of exchange `seq` stands in this rank's mailbox
__device__ __forceinline__ const unsigned long long* ll_red_word(const Ctx& c, uint32_t seq, int q, int i) {
  return c.xll_red + (((size_t)(seq & (kLLSlots - 1)) * c.xsize + q) * c.xcap + i) * 2;
}
// ... and value f of vector v of the ghost node with receive-list index k, as sent by exchange `seq`
__device__ __forceinline__ const unsigned long long* ll_ghost_word(const Ctx& c, uint32_t seq, int k, int v, int f, int nf) {
  return c.xll_halo + (((size_t)k * kLLSlots + (seq & (kLLSlots - 1))) * kLLRow + v * nf + f) * 2;
}
constexpr int kXRanksMax = 8;   // ranks of a partition over the peer transport (= kPeerMax)
// The all-reduce inside a coarse workgroup: NCON entries, thread (q, i) = q * NCON + i polls rank q's contribution to entry i —
// every rank's words are in flight together — and thread i adds them up in RANK ORDER (the same order, hence the same bits, on
// every rank and in every workgroup).  entry(i) -> (index into the exchange's sums, which exchange).  Two barriers.
template <int NCON, class F>
__device__ __forceinline__ void ll_allreduce(const Ctx& c, double* con /* [kXRanksMax * NCON] */, double* dst /* [NCON] */, F entry) {
  const int t = threadIdx.x;
  if (t < NCON * c.xsize) {
    const int q = t / NCON, i = t - q * NCON;
    int idx; uint32_t seq;
    entry(i, idx, seq);
    const unsigned long long* const src[1] = {ll_red_word(c, seq, q, idx)};
    const uint32_t sq[1] = {seq};
    double v[1];
    ll_wait_n<1>(c, src, sq, v);
    con[t] = v[0];
  }
  __syncthreads();
  if (t < NCON) {
    double acc = 0.0;
    for (int q = 0; q < c.xsize; ++q) acc += con[q * NCON + t];
    dst[t] = acc;
  }
}

// ---- coarse kernels: scalars of the half-iteration, yc = Aci * (P^T p  or  P^T s) in column blocks ----------------
// One workgroup per aggregate g.  It sums the per-tile restriction partials of ITS NF coarse dofs only (fixed order),
// forms the NF entries of the coarse operand and writes the column-block product yc[g][:] = Aci[:, g-block] * operand_g;
// the tile kernels add the nagg blocks for the few coarse dofs they prolong from (TileCoarse).  Every workgroup
// reduces the scalar partials redundantly; workgroup 0 publishes the scalars.
template <int NF, bool FUSED, bool XCH = false>
__device__ __forceinline__ void coarse_a_body(const Ctx& c, const int k, const int g, const unsigned target) {
  __shared__ double cs[4 * NF + 4];
  __shared__ double lred[(kCoarseThreads / 64) * 4];
  KrylovScalars* sc = c.scal;
  const int t = threadIdx.x, n = c.ncoarse;
  const int par = k & 1;  // the iteration index comes from the host: no load stands in front of the requests below
  const bool first = (k == 0);
  // all requests first (the `done` flag among them: a finished solve still issues them, then exits)
  GMPNP_STAMP(0);
  if (XCH && g == 0) GMPNP_XSTAMP(k, 16);
  const int done_flag = sc->done;
  double acol[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acol[f] = c.Aci[(size_t)min(t, n - 1) * n + g * NF + f];
  PartialSums<4> psum;
  // Exact restrictions of the ACTUAL fine vectors of iteration k-1 (epilogue partials of A(k-1) and B(k-1)): the
  // coarse vectors are rebuilt from them every iteration, nothing accumulates on the coarse level.
  AggSlotSums<4, NF> ss;
  if (!XCH) {   // (exchange-prologue launches get their sums out of the mailbox: the partials are the exchange workgroups' business)
    psum.load(c.part_b, c.ntiles, c.ntiles);  // k = 0: stale values, not used
    // P^T v_{k-1} (k = 0: P^T b from k_restrict), P^T t_{k-1}, P^T r_{k-1}, P^T p_{k-1}
    const double* const arr[4] = {c.cpart_v[par ^ 1], first ? c.cpart_v[par ^ 1] : c.cpart_t,
                                  first ? c.cpart_v[par ^ 1] : c.cpart_r[par ^ 1], first ? c.cpart_v[par ^ 1] : c.cpart_p[par ^ 1]};
    ss.load(arr, n, c.tile_slots, g);
  }
  const double sc_alpha = sc->alpha, sc_rho0 = sc->rho[0], sc_rho1 = sc->rho[1];
  { double keep = 0.0;
    if (!XCH) {
      keep = ss.r[0];
#pragma unroll
      for (int u = 1; u < 9; ++u) keep += ss.r[u];
#pragma unroll
      for (int q = 0; q < 4; ++q) keep += psum.r[q][0] + psum.r[q][1];
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) keep += acol[f];
    GMPNP_EXIT_IF_DONE(done_flag, keep); }
  double alpha = 0.0, rho_old = 1.0, rho_new = 0.0, omega = 0.0, beta = 0.0;
  if (first) rho_new = sc_rho1;  // the host puts (rhat, r_0) = ||b||^2 there
  else { alpha = sc_alpha; rho_old = par ? sc_rho0 : sc_rho1; }
  GMPNP_STAMP(1);
  // ONE barrier: restriction sums of this aggregate and per-wave scalar sums go to LDS, then every thread finishes alone
  if (c.dist) {  // partitioned solve: the sums over all ranks' tiles were all-reduced into red_i / red_a / red_b
    if (XCH && g == 0) GMPNP_XSTAMP(k, 17);
    if (XCH) {
      // half A in the exchange-prologue form: the sums of phase 2 (what B(k-1) left: (t,s) (t,t) (rhat,s) (rhat,t) | P^T t) are those of
      // THIS launch's exchange, the sums of phase 1 (P^T v, r, p of A(k-1)) those of the exchange before; 4 NF + 4 threads poll one each
      __shared__ double con[kXRanksMax * (4 * NF + 4)];
      ll_allreduce<4 * NF + 4>(c, con, cs, [&](int i, int& idx, uint32_t& seq) {
        const int which = i / NF, f = i - which * NF, d = g * NF + f;
        if (i >= 4 * NF) { idx = i - 4 * NF; seq = c.xseq; }
        else if (which == 1) { idx = 4 + d; seq = c.xseq; }
        else { idx = 2 + (which == 0 ? 0 : which == 2 ? n : 2 * n) + d; seq = c.xseq - 1u; }
      });
    } else if (c.use_coarse && t < 4 * NF) {
      const int which = t / NF, f = t - which * NF, d = g * NF + f;
      cs[t] = first ? c.red_i[d] : (which == 0 ? c.red_a[2 + d] : which == 1 ? c.red_b[4 + d] : which == 2 ? c.red_a[2 + n + d] : c.red_a[2 + 2 * n + d]);
    }
  } else {
    if (c.use_coarse) ss.to_lds(cs, n, c.tile_slots);
    if (!first) psum.reduce_partial(lred, c.part_b, c.ntiles, c.ntiles);
  }
  __syncthreads();
  GMPNP_STAMP(2);
  if (XCH && g == 0) GMPNP_XSTAMP(k, 18);
  if (!first) {
    double tot[4];
    if (c.dist) {
      if (XCH) { tot[0] = cs[4 * NF]; tot[1] = cs[4 * NF + 1]; tot[2] = cs[4 * NF + 2]; tot[3] = cs[4 * NF + 3]; }
      else { tot[0] = c.red_b[0]; tot[1] = c.red_b[1]; tot[2] = c.red_b[2]; tot[3] = c.red_b[3]; }
    }
    else PartialSums<4>::reduce_final(lred, tot);
    omega = tot[0] / tot[1];
    rho_new = tot[2] - omega * tot[3];
    beta = (rho_new / rho_old) * (alpha / omega);
  }
  GMPNP_STAMP(3);
  if (XCH && g == 0) GMPNP_XSTAMP(k, 19);
  if (g == 0 && t == 0) { store_coherent<FUSED>(&sc->omega, omega); store_coherent<FUSED>(&sc->beta, beta); sc->rho[par] = rho_new; }
  if (!c.use_coarse) { publish_ticket<FUSED>(c, target, g); return; }
  GMPNP_STAMP(4);
  if (t < n) {
    // yc = Aci[:, g-block] (P^T p_k)_g with P^T p_k = P^T r_k + beta (P^T p_{k-1} - omega P^T v), P^T r_k = P^T r_{k-1} -
    // alpha P^T v - omega P^T t: the four block products do not wait for the scalars, only their combination does
    double yv = 0.0, yt = 0.0, yr = 0.0, yp = 0.0;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      yv += acol[f] * cs[f]; yt += acol[f] * cs[NF + f]; yr += acol[f] * cs[2 * NF + f]; yp += acol[f] * cs[3 * NF + f];
    }
    const double acc = first ? yv : ((yr - alpha * yv) - omega * yt) + beta * (yp - omega * yv);
    store_coherent<FUSED>(&c.yc[(size_t)g * n + t], acc);
  }
  publish_ticket<FUSED>(c, target, g);
  GMPNP_STAMP(5);
  if (XCH && g == 0) GMPNP_XSTAMP(k, 20);
}

template <int NF, bool FUSED, bool XCH = false>
__device__ __forceinline__ void coarse_b_body(const Ctx& c, const int k, const int g, const unsigned target) {
  __shared__ double cs[2 * NF + 2];
  __shared__ double lred[(kCoarseThreads / 64) * 2];
  KrylovScalars* sc = c.scal;
  const int t = threadIdx.x, n = c.ncoarse;
  const int par = k & 1;
  const int done_flag = sc->done, max_iters = sc->max_iters;
  const double rho_new = sc->rho[par], tol = sc->tol, rr0 = sc->rr0;
  double acol[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acol[f] = c.Aci[(size_t)min(t, n - 1) * n + g * NF + f];
  PartialSums<2> psum;  // part_a and part_rr are adjacent halves of one buffer: stride ntiles
  AggSlotSums<2, NF> ss;
  if (!XCH) { psum.load(c.part_a, c.ntiles, c.ntiles); const double* const arr[2] = {c.cpart_v[par], c.cpart_r[par]}; ss.load(arr, n, c.tile_slots, g); }
  { double keep = 0.0;
    if (!XCH) {
      keep = ss.r[0] + psum.r[0][0] + psum.r[1][0];
#pragma unroll
      for (int u = 1; u < 9; ++u) keep += ss.r[u];
      keep += (psum.r[0][1] + psum.r[1][1]);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) keep += acol[f];
    GMPNP_EXIT_IF_DONE(done_flag, keep); }
  if (c.dist) {
    if (XCH) {   // half B: the sums of phase 1 ((rhat,v) ||r||^2 | P^T v | P^T r | ...) are those of THIS launch's exchange
      __shared__ double con[kXRanksMax * (2 * NF + 2)];
      ll_allreduce<2 * NF + 2>(c, con, cs, [&](int i, int& idx, uint32_t& seq) {
        const int which = i / NF, f = i - which * NF;
        idx = i >= 2 * NF ? i - 2 * NF : 2 + which * n + g * NF + f; seq = c.xseq;
      });
    } else if (c.use_coarse && t < 2 * NF) { const int which = t / NF, f = t - which * NF; cs[t] = c.red_a[2 + which * n + g * NF + f]; }
  } else {
    if (c.use_coarse) ss.to_lds(cs, n, c.tile_slots);
    psum.reduce_partial(lred, c.part_a, c.ntiles, c.ntiles);
  }
  __syncthreads();
  double tot[2];
  if (c.dist) { tot[0] = XCH ? cs[2 * NF] : c.red_a[0]; tot[1] = XCH ? cs[2 * NF + 1] : c.red_a[1]; }
  else PartialSums<2>::reduce_final(lred, tot);
  const double rv = tot[0], rr = tot[1];
  int done = 0;
  if (!(rr == rr) || !(rv == rv)) done = 3;
  else if (sqrt(rr) <= tol) done = 1;
  else if (rr > 1e10 * rr0) done = 3;  // residual 1e5 times its start: this pass is lost (BiCGStab spikes stay far below)
  else if (k >= max_iters) done = 2;
  else if (rv == 0.0 || rho_new == 0.0) done = 3;
  // `done` is published by the B kernel (the launch after this one): other workgroups of THIS launch still read it
  const double alpha = done ? 0.0 : rho_new / rv;
  if (g == 0 && t == 0) { store_coherent<FUSED>(&sc->alpha, alpha); store_coherent<FUSED>(&sc->rr, rr); store_coherent<FUSED>(&sc->done_next, (int32_t)done); }
  if (done || !c.use_coarse) { publish_ticket<FUSED>(c, target, g); return; }
  if (t < n) {  // yc = Aci[:, g-block] (P^T s)_g, P^T s = P^T r_k - alpha P^T v_k
    double yv = 0.0, yr = 0.0;
#pragma unroll
    for (int f = 0; f < NF; ++f) { yv += acol[f] * cs[f]; yr += acol[f] * cs[NF + f]; }
    store_coherent<FUSED>(&c.yc[(size_t)g * n + t], yr - alpha * yv);
  }
  publish_ticket<FUSED>(c, target, g);
}

// Results of a tile kernel (its rows of the new vectors, its partial sums) are read by the NEXT launch, on other XCDs as well:
// stored write-through, so that the write-back of the XCD's L2 at the end of the kernel finds nothing left to do (k_half_a
// 10.68 -> 10.52 us, k_half_b 9.95 -> 9.66 us; non-temporal stores instead: slower than plain ones).
__device__ __forceinline__ void st_out(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ---- fused half-iterations ------------------------------------------------------------------------------------
// MAT ("materialised"): p_k (resp. s_k) was written for ALL rows by k_vec_a (k_vec_b) in front of this launch, together with
// the own-row updates of y and r; the tile stages ONE vector at its column nodes instead of recomputing p from four (s from
// two).  Pays where the operand gathers cost HBM / Infinity-Cache bandwidth (twice-refined meshes: a tile's four staged vectors
// are as many bytes as its matrix slice), not on the reference meshes, where one more launch per half-iteration costs more.
// XCH (partitioned solve over the peer transport, exchange workgroups in front of this launch's coarse ones): the ghost entries of
// the staged vectors are not read from the vectors (whose ghost rows nobody keeps current in this form) but out of the mailbox, as
// flagged words of this launch's exchange and of the one before (ll_ghost), behind the hand-over; boundary tiles only.
template <int NF, bool FUSED, bool MAT = false, bool XCH = false>
__device__ __forceinline__ void bicg_a_body(const Ctx& c, const int k, const int tile, const unsigned target) {
  static_assert(!(FUSED && MAT), "the materialised form has its own launches");
  static_assert(!XCH || FUSED, "the exchange prologue belongs to the fused launch form");
  constexpr int NW = kKrylovWaves;
  __shared__ double red[kSlicesPerTile * NW][64];
  __shared__ double outv[3][kSlicesPerTile][64];  // v, r, p of the tile's rows
  __shared__ double dpart[kSlicesPerTile][2];
  __shared__ double xs[kTileCols * NF];
  __shared__ double xt[MAT ? 1 : kTileCols * NF];  // s and t of the staged entries wait in LDS (xs, xt), not in registers
  __shared__ double ycl[kMaxCoarse];
  __shared__ double own[6][64];
  static_assert(kSlicesPerTile == 1 && NW >= 6, "own-row hand-over: one slice per tile, one wave per vector");
  KrylovScalars* sc = c.scal;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int sl = wv / NW;
  const int par = k & 1, n = c.ncoarse;
  const bool first = (k == 0);
  GMPNP_STAMP(0);
  if (XCH && tile == c.tile0) GMPNP_XSTAMP(k, 24);
  const int done_flag = sc->done;
  const double alpha = sc->alpha;
  double omega = 0.0, beta = 0.0;
  if (!FUSED) { omega = sc->omega; beta = sc->beta; }
  const double* __restrict__ po = c.kp[par ^ 1];
  const double* __restrict__ vo = c.kv[par ^ 1];
  // materialised form: the staged operand is p_k itself, or p_k + theta D T p_k when a multilevel term is attached (gmpnp_multilevel.h)
  const double* __restrict__ matsrc = (MAT && c.stage_a) ? c.stage_a : c.kp[par];
  const double* __restrict__ sfirst = MAT ? matsrc : (first ? c.kr : c.ks);  // k = 0: s, t, p_old, v_old do not exist yet, p_0 = r_0
  // every global request of this launch, issued together
  // two independent chains: the tile record -> matrix values / local column indices, and the tile's column list
  // (fixed stride: addressable without the record) -> operands of the staged x entries
  const TileRec rec = c.tile_rec[tile];
  const int c0 = tile * c.col_stride;
  int st_col[kStagePre], st_agg[kStagePre];
#pragma unroll
  for (int u = 0; u < kStagePre; ++u) {
    const int cl = min((t + u * kKrylovThreads) / NF, c.col_stride - 1);  // padding entries of the list point at node 0
    st_col[u] = (XCH ? c.tile_cols_x : c.tile_cols)[c0 + cl]; st_agg[u] = c.tile_colslot[c0 + cl];   // XCH: ghost nodes are -(receive-list index + 1)
  }
  TileCoarse<NF> tcs;
  tcs.load_index(c, tile);
  if (!FUSED) tcs.template load_values<false>(c);
  TileRows<NF, (FUSED ? GMPNP_ROW_PRELOAD_A : GMPNP_ROW_PRELOAD)> rows;
  rows.load(c, c.vals_s, rec);
  const int nst = rec.ncols * NF;
  double st_s[kStagePre], st_t[kStagePre], st_p[kStagePre], st_v[kStagePre];
#pragma unroll
  for (int u = 0; u < kStagePre; ++u) {
    const int q = t + u * kKrylovThreads;
    const bool gh = XCH && st_col[u] < 0;
    // XCH: a staged GHOST entry (its values come out of the mailbox) keeps its receive-list index + 1 above the slot number in st_agg
    if (XCH && gh && q < nst) st_agg[u] |= -st_col[u] << 8;
    // (a ghost entry requests an owned row instead: the vectors' ghost rows are not kept up to date in this form)
    const unsigned off = ((unsigned)(gh ? c.own_node0 : st_col[u]) * NF + (unsigned)(q - (q / NF) * NF)) * 8u;
    st_s[u] = ld_off(sfirst, off);
    if (MAT) { st_t[u] = 0.0; st_p[u] = 0.0; st_v[u] = 0.0; }
    else { st_t[u] = ld_off(c.kt, off); st_p[u] = ld_off(po, off); st_v[u] = ld_off(vo, off); }  // k = 0: only st_s is used
  }
  // own rows (the epilogue of wave 0 needs six vectors at its rows): wave q requests vector q and hands it over through
  // LDS, so that no load sits behind a branch and nobody holds six values
  const int own_r = rows.row;  // inactive lanes: row 0
  const double* ownp = MAT ? (wv == 0 ? c.krhat : wv == 1 ? c.kr : c.kp[par])   // r_k and p_k are in place already
                           : (wv == 0 ? c.krhat : wv == 1 ? sfirst : wv == 2 ? c.kt : wv == 3 ? po : wv == 4 ? vo : c.ky);
  const double own_q = ownp[own_r];
  // The staged s and t go to LDS as soon as they arrive (they were requested first, so this wait does not cover the matrix
  // preload): eight registers less behind the hand-over, which is what lets the fused A half preload a second block position
  // without spilling.  p_old and v_old stay in registers.
  if (!MAT && !first) {
#pragma unroll
    for (int u = 0; u < kStagePre; ++u) { xs[t + u * kKrylovThreads] = st_s[u]; xt[t + u * kKrylovThreads] = st_t[u]; }
  }
  { double keep = own_q + (FUSED ? 0.0 : tcs.a0 + tcs.a1);
#pragma unroll
    for (int u = 0; u < kStagePre; ++u) keep += ((MAT || first) ? st_s[u] + st_t[u] : 0.0) + (st_p[u] + st_v[u]);
#pragma unroll
    for (int u = 0; u < decltype(rows)::PRE; ++u)
#pragma unroll
      for (int j = 0; j < NF; ++j) keep += rows.av[u][j];
    GMPNP_EXIT_IF_DONE(done_flag, keep); }
  if (XCH && tile == c.tile0) GMPNP_XSTAMP(k, 25);
  if (FUSED) {  // scalars and coarse products of THIS launch's coarse workgroups
    if (!wait_ticket(c, target, XCH ? kXchBudget + 200000000ull : 200000000ull)) return;
    if (XCH && tile == c.tile0) GMPNP_XSTAMP(k, 26);
    // the two scalars: ONE wave asks (one request instead of eight per tile on the line every tile reads at this moment), the
    // others get them through LDS behind the barrier that follows anyway
    __shared__ double hand[2];
    double w0 = 0.0, b0 = 0.0;
    if (wv == 0) { w0 = load_coherent(&sc->omega); b0 = load_coherent(&sc->beta); }
    tcs.template load_values<true>(c);
    if (XCH) {   // ghost entries (boundary tiles only): s and t as sent by THIS launch's exchange, p_old and v_old by the one before
#pragma unroll
      for (int u = 0; u < kStagePre; ++u)
        if (st_agg[u] >> 8) {
          const int q = t + u * kKrylovThreads, f = q - (q / NF) * NF, gk = (st_agg[u] >> 8) - 1;
          const unsigned long long* const src[4] = {ll_ghost_word(c, c.xseq, gk, 0, f, NF), ll_ghost_word(c, c.xseq, gk, 1, f, NF),
                                                    ll_ghost_word(c, c.xseq - 1u, gk, 1, f, NF), ll_ghost_word(c, c.xseq - 1u, gk, 2, f, NF)};
          const uint32_t sq[4] = {c.xseq, c.xseq, c.xseq - 1u, c.xseq - 1u};
          double g4[4];
          ll_wait_n<4>(c, src, sq, g4);
          xs[q] = g4[0]; xt[q] = g4[1]; st_v[u] = g4[2]; st_p[u] = g4[3];
        }
    }
    if (t == 0) { hand[0] = w0; hand[1] = b0; }
    if (c.use_coarse) tcs.to_lds(c, ycl);
    __syncthreads();
    omega = hand[0]; beta = hand[1];
  }
  GMPNP_STAMP(1);
  const bool uc = c.use_coarse != 0;
  if (!FUSED && uc) { tcs.to_lds(c, ycl); __syncthreads(); }
  GMPNP_STAMP(2);
  // stage x = p_new + P yc for the tile's column nodes
#pragma unroll
  for (int u = 0; u < kStagePre; ++u) {
    const int q = t + u * kKrylovThreads;
    if (q < nst) {
      const double pj = (first || MAT) ? st_s[u] : (xs[q] - omega * xt[q]) + beta * (st_p[u] - omega * st_v[u]);
      xs[q] = pj + (uc ? ycl[(XCH ? st_agg[u] & 0xff : st_agg[u]) * NF + (q - (q / NF) * NF)] : 0.0);
    }
  }
  if (wv < 6) own[wv][lane] = own_q;
  for (int q = t + kStagePre * kKrylovThreads; q < nst; q += kKrylovThreads) {
    const int cl = q / NF, f = q - cl * NF;
    const int node = (XCH ? c.tile_cols_x : c.tile_cols)[c0 + cl];
    double pj;
    if (XCH && node < 0) {
      const int gk = -node - 1;
      const unsigned long long* const src[4] = {ll_ghost_word(c, c.xseq, gk, 0, f, NF), ll_ghost_word(c, c.xseq, gk, 1, f, NF),
                                                ll_ghost_word(c, c.xseq - 1u, gk, 1, f, NF), ll_ghost_word(c, c.xseq - 1u, gk, 2, f, NF)};
      const uint32_t sq[4] = {c.xseq, c.xseq, c.xseq - 1u, c.xseq - 1u};
      double g4[4];
      ll_wait_n<4>(c, src, sq, g4);
      pj = (g4[0] - omega * g4[1]) + beta * (g4[3] - omega * g4[2]);
    } else {
      const size_t idx = (size_t)node * NF + f;
      pj = MAT ? matsrc[idx] : first ? c.kr[idx] : (c.ks[idx] - omega * c.kt[idx]) + beta * (po[idx] - omega * vo[idx]);
    }
    xs[q] = pj + (uc ? ycl[c.tile_colslot[c0 + cl] * NF + f] : 0.0);
  }
  __syncthreads();
  GMPNP_STAMP(3);
  if (XCH && tile == c.tile0) GMPNP_XSTAMP(k, 28);
  red[wv][lane] = rows.dot(c, xs);
  __syncthreads();
  GMPNP_STAMP(4);
  // own rows, dots, partial restriction
  if (rows.w == 0) {
    double tot = 0.0, rn = 0.0, pn = 0.0;
    const double own_rh = own[0][lane], own_s = own[1][lane], own_t = own[2][lane], own_p = own[3][lane], own_v = own[4][lane], own_y = own[5][lane];
    if (rows.active) {
#pragma unroll
      for (int q = 0; q < NW; ++q) tot += red[sl * NW + q][lane];
      if (MAT) { rn = own_s; pn = own_t; }   // own[1] = r_k, own[2] = p_k (k_vec_a)
      else {
        if (first) { rn = own_s; pn = rn; }
        else {
          st_out(&c.ky[own_r], own_y + alpha * own_p + omega * own_s);
          rn = own_s - omega * own_t;
          st_out(&c.kr[own_r], rn);
          pn = rn + beta * (own_p - omega * own_v);
        }
        st_out(&c.kp[par][own_r], pn);
      }
      st_out(&c.kv[par][own_r], tot);
    }
    outv[0][sl][lane] = tot; outv[1][sl][lane] = rn; outv[2][sl][lane] = pn;
    const double d0 = wave_sum(own_rh * tot), d1 = wave_sum(rn * rn);
    if (lane == 0) { dpart[sl][0] = d0; dpart[sl][1] = d1; }
  }
  __syncthreads();
  if (t < 3 * NF) {  // partial restrictions of v, r, p over the tile's rows (all in aggregate tile_agg)
    const int which = t / NF, f = t - which * NF;
    double sacc = 0.0;
    for (int q = 0; q < kSlicesPerTile; ++q)
      for (int il = 0; il < kWave / NF; ++il) sacc += outv[which][q][il * NF + f];
    double* dstp = which == 0 ? c.cpart_v[par] : (which == 1 ? c.cpart_r[par] : c.cpart_p[par]);
    st_out(&dstp[cpart_index(c, rec.slot, rec.agg, f, NF)], sacc);
  } else if (t == 64) {
    double a0 = 0.0, a1 = 0.0;
    for (int q = 0; q < kSlicesPerTile; ++q) { a0 += dpart[q][0]; a1 += dpart[q][1]; }
    st_out(&c.part_a[tile], a0); st_out(&c.part_rr[tile], a1);
  }
  if (XCH && tile == c.tile0) { __syncthreads(); GMPNP_XSTAMP(k, 27); }
#ifdef GMPNP_TIMING
  __syncthreads();
  GMPNP_STAMP(5);
#endif
}

template <int NF, bool FUSED, bool MAT = false, bool XCH = false>
__device__ __forceinline__ void bicg_b_body(const Ctx& c, const int k, const int tile, const unsigned target) {
  static_assert(!(FUSED && MAT), "the materialised form has its own launches");
  static_assert(!XCH || FUSED, "the exchange prologue belongs to the fused launch form");
  constexpr int NW = kKrylovWaves;
  __shared__ double red[kSlicesPerTile * NW][64];
  __shared__ double outv[kSlicesPerTile][64];
  __shared__ double dpart[kSlicesPerTile][4];
  __shared__ double xs[kTileCols * NF];
  __shared__ double ycl[kMaxCoarse];
  __shared__ double own[3][64];
  KrylovScalars* sc = c.scal;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int sl = wv / NW;
  const int done_flag = sc->done;
  int dn = 0;  // verdict of k_coarse_b(k) on ||r_k||
  const int par = k & 1, n = c.ncoarse;
  double alpha = 0.0;
  if (!FUSED) { dn = sc->done_next; alpha = sc->alpha; }
  const double* __restrict__ vn = c.kv[par];
  const TileRec rec = c.tile_rec[tile];
  const int c0 = tile * c.col_stride;
  int st_col[kStagePre], st_agg[kStagePre];
#pragma unroll
  for (int u = 0; u < kStagePre; ++u) {
    const int cl = min((t + u * kKrylovThreads) / NF, c.col_stride - 1);  // padding entries of the list point at node 0
    st_col[u] = (XCH ? c.tile_cols_x : c.tile_cols)[c0 + cl]; st_agg[u] = c.tile_colslot[c0 + cl];
  }
  TileCoarse<NF> tcs;
  tcs.load_index(c, tile);
  if (!FUSED) tcs.template load_values<false>(c);
  TileRows<NF, (FUSED ? GMPNP_ROW_PRELOAD_B : GMPNP_ROW_PRELOAD)> rows;
  rows.load(c, c.vals_s, rec);
  const int nst = rec.ncols * NF;
  double st_r[kStagePre], st_v[kStagePre];
#pragma unroll
  for (int u = 0; u < kStagePre; ++u) {
    const int q = t + u * kKrylovThreads;
    const bool gh = XCH && st_col[u] < 0;
    if (XCH && gh && q < nst) st_agg[u] |= -st_col[u] << 8;   // see bicg_a_body
    const unsigned off = ((unsigned)(gh ? c.own_node0 : st_col[u]) * NF + (unsigned)(q - (q / NF) * NF)) * 8u;
    st_r[u] = ld_off(MAT ? ((MAT && c.stage_b) ? c.stage_b : c.ks) : c.kr, off); st_v[u] = MAT ? 0.0 : ld_off(vn, off);   // MAT: s_k was written by k_vec_b (+ the multilevel term)
  }
  const int own_r = rows.row;  // inactive lanes: row 0
  const double* ownp = wv == 0 ? c.krhat : wv == 1 ? (MAT ? c.ks : c.kr) : vn;  // wave q requests own-row vector q (see k_bicg_a)
  const double own_q = ownp[own_r];
  { double keep = own_q + (FUSED ? 0.0 : tcs.a0 + tcs.a1);
#pragma unroll
    for (int u = 0; u < kStagePre; ++u) keep += st_r[u] + st_v[u];
#pragma unroll
    for (int u = 0; u < decltype(rows)::PRE; ++u)
#pragma unroll
      for (int j = 0; j < NF; ++j) keep += rows.av[u][j];
    GMPNP_EXIT_IF_DONE(done_flag, keep); }
  if (FUSED) {
    if (!wait_ticket(c, target, XCH ? kXchBudget + 200000000ull : 200000000ull)) return;
    __shared__ double hand[2];   // see bicg_a_body
    double d0 = 0.0, a0 = 0.0;
    if (wv == 0) { d0 = load_coherent(&sc->done_next); a0 = load_coherent(&sc->alpha); }
    tcs.template load_values<true>(c);
    if (XCH) {   // ghost entries (boundary tiles only): r and v as sent by THIS launch's exchange
#pragma unroll
      for (int u = 0; u < kStagePre; ++u)
        if (st_agg[u] >> 8) {
          const int q = t + u * kKrylovThreads, f = q - (q / NF) * NF, gk = (st_agg[u] >> 8) - 1;
          const unsigned long long* const src[2] = {ll_ghost_word(c, c.xseq, gk, 0, f, NF), ll_ghost_word(c, c.xseq, gk, 1, f, NF)};
          const uint32_t sq[2] = {c.xseq, c.xseq};
          double g2[2];
          ll_wait_n<2>(c, src, sq, g2);
          st_r[u] = g2[0]; st_v[u] = g2[1];
        }
    }
    if (t == 0) { hand[0] = d0; hand[1] = a0; }
    if (c.use_coarse) tcs.to_lds(c, ycl);
    __syncthreads();
    dn = (int)hand[0]; alpha = hand[1];
  }
  if (dn) {
    if (tile == c.tile0 && t == 0) {
      sc->done = dn;  // published here: no workgroup of THIS launch reads it any more... others exit on dn
      poll_finish(c, FUSED ? load_coherent(&sc->rr) : sc->rr, k, dn);  // the verdict of coarse_b(k) is on r_k: k iterations done
    }
    return;
  }
  const bool uc = c.use_coarse != 0;
  if (!FUSED && uc) { tcs.to_lds(c, ycl); __syncthreads(); }
#pragma unroll
  for (int u = 0; u < kStagePre; ++u) {
    const int q = t + u * kKrylovThreads;
    if (q < nst) xs[q] = (MAT ? st_r[u] : st_r[u] - alpha * st_v[u]) + (uc ? ycl[(XCH ? st_agg[u] & 0xff : st_agg[u]) * NF + (q - (q / NF) * NF)] : 0.0);
  }
  if (wv < 3) own[wv][lane] = own_q;
  for (int q = t + kStagePre * kKrylovThreads; q < nst; q += kKrylovThreads) {
    const int cl = q / NF, f = q - cl * NF;
    const int node = (XCH ? c.tile_cols_x : c.tile_cols)[c0 + cl];
    double sj;
    if (XCH && node < 0) {
      const unsigned long long* const src[2] = {ll_ghost_word(c, c.xseq, -node - 1, 0, f, NF), ll_ghost_word(c, c.xseq, -node - 1, 1, f, NF)};
      const uint32_t sq[2] = {c.xseq, c.xseq};
      double g2[2];
      ll_wait_n<2>(c, src, sq, g2);
      sj = g2[0] - alpha * g2[1];
    } else {
      const size_t idx = (size_t)node * NF + f;
      sj = MAT ? ((MAT && c.stage_b) ? c.stage_b : c.ks)[idx] : c.kr[idx] - alpha * vn[idx];
    }
    xs[q] = sj + (uc ? ycl[c.tile_colslot[c0 + cl] * NF + f] : 0.0);
  }
  __syncthreads();
  red[wv][lane] = rows.dot(c, xs);
  __syncthreads();
  if (rows.w == 0) {
    double tt = 0.0, sv = 0.0;
    const double own_rh = own[0][lane], own_r_ = own[1][lane], own_v = own[2][lane];
    if (rows.active) {
#pragma unroll
      for (int q = 0; q < NW; ++q) tt += red[sl * NW + q][lane];
      sv = MAT ? own_r_ : own_r_ - alpha * own_v;
      if (!MAT) st_out(&c.ks[own_r], sv);
      st_out(&c.kt[own_r], tt);
    }
    outv[sl][lane] = tt;
    const double d0 = wave_sum(tt * sv), d1 = wave_sum(tt * tt), d2 = wave_sum(own_rh * sv), d3 = wave_sum(own_rh * tt);
    if (lane == 0) { dpart[sl][0] = d0; dpart[sl][1] = d1; dpart[sl][2] = d2; dpart[sl][3] = d3; }
  }
  __syncthreads();
  if (t < NF) {
    double sacc = 0.0;
    for (int q = 0; q < kSlicesPerTile; ++q)
      for (int il = 0; il < kWave / NF; ++il) sacc += outv[q][il * NF + t];
    st_out(&c.cpart_t[cpart_index(c, rec.slot, rec.agg, t, NF)], sacc);
  } else if (t == 64) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      double a0 = 0.0;
      for (int q = 0; q < kSlicesPerTile; ++q) a0 += dpart[q][m];
      st_out(&c.part_b[(size_t)m * c.ntiles + tile], a0);
    }
    if (tile == c.tile0) { sc->iters = k + 1; poll_store(&c.poll->iters, k + 1); }
  }
}

// ---- block index -> tile -----------------------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (blocks q and q + 8 share one; observed, relied on for speed only), and an
// XCD's L2 serves every tile that runs there during the launch.  Tiles in mesh order are neighbours along the pore axis and
// stage largely the same column nodes, so XCD x gets a CONTIGUOUS eighth of the tiles: each staged vector line then
// crosses the fabric once per launch instead of once per XCD.  Sums are indexed by tile, not by block: results do not change.
#ifndef GMPNP_XCD_TILES
#define GMPNP_XCD_TILES 1
#endif
__device__ __forceinline__ int xcd_tile(int q, int ntiles) {
  if (!GMPNP_XCD_TILES) return q;
  const int x = q & (kXcds - 1), i = q >> 3, per = ntiles >> 3, rem = ntiles & (kXcds - 1);
  return x * per + min(x, rem) + i;
}

// ---- launch forms ---------------------------------------------------------------------------------------------
// Four launches per iteration (coarse_a, bicg_a, coarse_b, bicg_b) ...
template <int NF>
__global__ __launch_bounds__(kCoarseThreads) void k_coarse_a(const Ctx c, const int k) { coarse_a_body<NF, false>(c, k, blockIdx.x, 0u); }
template <int NF>
__global__ __launch_bounds__(kCoarseThreads) void k_coarse_b(const Ctx c, const int k) { coarse_b_body<NF, false>(c, k, blockIdx.x, 0u); }
template <int NF>
__global__ __launch_bounds__(kKrylovThreads) void k_bicg_a(const Ctx c, const int k) { bicg_a_body<NF, false>(c, k, c.tile0 + xcd_tile(blockIdx.x, gridDim.x), 0u); }
template <int NF>
__global__ __launch_bounds__(kKrylovThreads) void k_bicg_b(const Ctx c, const int k) { bicg_b_body<NF, false>(c, k, c.tile0 + xcd_tile(blockIdx.x, gridDim.x), 0u); }

// Materialised form (large meshes): the vector recurrences run as their own streaming launches in front of the tile kernels.
template <int NF>
__global__ __launch_bounds__(kKrylovThreads) void k_bicg_a_mat(const Ctx c, const int k) { bicg_a_body<NF, false, true>(c, k, c.tile0 + xcd_tile(blockIdx.x, gridDim.x), 0u); }
template <int NF>
__global__ __launch_bounds__(kKrylovThreads) void k_bicg_b_mat(const Ctx c, const int k) { bicg_b_body<NF, false, true>(c, k, c.tile0 + xcd_tile(blockIdx.x, gridDim.x), 0u); }
// y += alpha p_{k-1} + omega s ; r_k = s - omega t ; p_k = r_k + beta (p_{k-1} - omega v_{k-1})   (k = 0: p_0 = r_0), all rows
__global__ __launch_bounds__(256) void k_vec_a(const Ctx c, const int k) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const KrylovScalars* sc = c.scal;
  if (i >= c.ndof || sc->done) return;
  const int par = k & 1;
  if (k == 0) { c.kp[par][i] = c.kr[i]; return; }
  const double alpha = sc->alpha, omega = sc->omega, beta = sc->beta;
  const double s = c.ks[i], t = c.kt[i], po = c.kp[par ^ 1][i], vo = c.kv[par ^ 1][i];
  c.ky[i] += alpha * po + omega * s;
  const double rn = s - omega * t;
  c.kr[i] = rn;
  c.kp[par][i] = rn + beta * (po - omega * vo);
}
// s_k = r_k - alpha v_k, all rows (nothing to do once coarse_b(k) has seen the end of the solve)
__global__ __launch_bounds__(256) void k_vec_b(const Ctx c, const int k) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const KrylovScalars* sc = c.scal;
  if (i >= c.ndof || sc->done || sc->done_next) return;
  c.ks[i] = c.kr[i] - sc->alpha * c.kv[k & 1][i];
}

// ... or two: the nagg coarse workgroups ride in front of the tile workgroups of the same launch.  A tile workgroup
// requests everything that does not depend on the coarse result (indices, matrix slice, operand vectors), then waits
// for the flags of all coarse workgroups to reach `target` (= launches so far in this solve) before it reads the scalars and
// the coarse products.  Coarse workgroups have the lowest block indices (dispatched first) and wait for nobody, so
// the flags always arrive; the wait is bounded by a wall-clock budget all the same (status bit 8).
static_assert(kCoarseThreads == kKrylovThreads, "coarse and tile workgroups share a launch");
template <int NF>
__global__ __launch_bounds__(kKrylovThreads, 6) void k_half_a(const Ctx c, const int k, const unsigned target) {
  if ((int)blockIdx.x < c.nagg) coarse_a_body<NF, true>(c, k, blockIdx.x, target);
  else bicg_a_body<NF, true>(c, k, c.tile0 + xcd_tile(blockIdx.x - c.nagg, gridDim.x - c.nagg), target);
}
template <int NF>
__global__ __launch_bounds__(kKrylovThreads, 6) void k_half_b(const Ctx c, const int k, const unsigned target) {
  if ((int)blockIdx.x < c.nagg) coarse_b_body<NF, true>(c, k, blockIdx.x, target);
  else bicg_b_body<NF, true>(c, k, c.tile0 + xcd_tile(blockIdx.x - c.nagg, gridDim.x - c.nagg), target);
}

// Plain y = A x with the UNSCALED matrix (parity hook, partitioned driver); same tiling as the Krylov kernels.
template <int NF, bool RES>
__device__ __forceinline__ void spmv_plain_body(const Ctx& c, const double* __restrict__ x, const double* __restrict__ b, double* __restrict__ out) {
  constexpr int NW = kKrylovWaves;
  __shared__ double red[kSlicesPerTile * NW][64];
  __shared__ double xs[kTileCols * NF];
  const int tile = c.tile0 + blockIdx.x, t = threadIdx.x, wv = t >> 6, lane = t & 63;   // owned tiles only
  const int sl = wv / NW;
  const TileRec rec = c.tile_rec[tile];
  const int c0 = tile * c.col_stride;
  for (int q = t; q < c.col_stride * NF; q += kKrylovThreads) {
    const int cl = q / NF, f = q - cl * NF;
    xs[q] = x[(size_t)c.tile_cols[c0 + cl] * NF + f];
  }
  TileRows<NF> rows;
  rows.load(c, c.vals, rec);
  __syncthreads();
  red[wv][lane] = rows.dot(c, xs);
  __syncthreads();
  if (rows.w != 0 || !rows.active) return;
  double tot = 0.0;
#pragma unroll
  for (int q = 0; q < NW; ++q) tot += red[sl * NW + q][lane];
  out[rows.row] = RES ? b[rows.row] - tot : tot;
}
template <int NF>
__global__ __launch_bounds__(kKrylovThreads) void k_spmv_plain(const Ctx c, const double* __restrict__ x, double* __restrict__ out) {
  spmv_plain_body<NF, false>(c, x, nullptr, out);
}
// out = b - A x in one launch (the smoothing steps of the multilevel term, gmpnp_multilevel.h)
template <int NF>
__global__ __launch_bounds__(kKrylovThreads) void k_spmv_residual(const Ctx c, const double* __restrict__ x, const double* __restrict__ b, double* __restrict__ out) {
  spmv_plain_body<NF, true>(c, x, b, out);
}

// As = A Dinv: one wave per (slice, block position) scales the NF-entry row pieces of its block by Dinv of the column node.
template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_scale_columns(const Ctx c) {
  const int wave = xcd_run_wave(c), lane = threadIdx.x & 63;
  if (wave >= c.n_work) return;
  const int s = c.wl_slice[wave], kpos = c.wl_kpos[wave];
  if (s < 0) return;   // padding of a run
  const int Iloc = lane / NF;
  if (Iloc >= c.slice_nn[s]) return;
  const size_t rec = (size_t)(c.slice_colbase[s] + kpos) * kSlicePad + Iloc;
  const size_t off = c.slice_off[s] + (size_t)kpos * NF * kWave + lane;
  if (c.sell_blk[rec] < 0) return;  // padding stays zero in vals_s as well
  const double* d = c.Dinv + (size_t)(c.sell_cols[rec] & 0xFFFFFF) * NF * NF;
  double a[NF], o[NF];
#pragma unroll
  for (int mI = 0; mI < NF; ++mI) a[mI] = c.vals[off + (size_t)mI * kWave];
#pragma unroll
  for (int j = 0; j < NF; ++j) o[j] = 0.0;
#pragma unroll
  for (int mI = 0; mI < NF; ++mI)
#pragma unroll
    for (int j = 0; j < NF; ++j) o[j] += a[mI] * d[mI * NF + j];
#pragma unroll
  for (int j = 0; j < NF; ++j) c.vals_s[off + (size_t)j * kWave] = o[j];
}

// Partial restriction of a fine vector per tile -> cpart_v (used for P^T b at the start and P^T y at the end).
template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_restrict(const Ctx c, const double* __restrict__ x, double* __restrict__ part) {
  __shared__ double lv[kSlicesPerTile * 64];
  const int tile = c.tile0 + blockIdx.x, t = threadIdx.x;
  if (t < kSlicesPerTile * 64) {
    const int sl = t >> 6, lane = t & 63, s = c.tile_slice0[tile] + sl;
    const int Iloc = lane / NF, i = lane - Iloc * NF;
    const bool active = s < c.tile_slice0[tile + 1] && Iloc < c.slice_nn[s];
    lv[t] = active ? x[(c.slice_node0[s] + Iloc) * NF + i] : 0.0;
  }
  __syncthreads();
  if (t < NF) {
    double sacc = 0.0;
    for (int q = 0; q < kSlicesPerTile; ++q)
      for (int il = 0; il < kWave / NF; ++il) sacc += lv[q * 64 + il * NF + t];
    part[cpart_index(c, c.tile_slot[tile], c.tile_agg[tile], t, NF)] = sacc;
  }
}

// dst = scale_dst*dst + scale_x * Dinv (x + P Aci P^T x), P^T x taken from the partials k_restrict left in `part`.
// `upd` (Newton, final application of a solve): in the same pass u -= omega dx and the predicted start of the next
// linear solve, dst <- a dx + b xp, xp <- dx  (see newton(): a = (1-w) + (1-w)^2, b = -(1-w)^3 from the third iteration on)
struct NewtonUpdate { double* u; double* xp; double omega, a, b; };
template <int NF>
__global__ __launch_bounds__(kKrylovThreads) void k_minv_apply(const Ctx c, const double* __restrict__ x, const double* __restrict__ part,
                                                                double* __restrict__ dst, double scale_dst, double scale_x,
                                                                const NewtonUpdate upd, const double* __restrict__ reduced) {
  __shared__ double pcs[kMaxCoarse];
  __shared__ double ycl[kTileAggs * NF];
  __shared__ double xv[kSlicesPerTile][64];
  const int tile = c.tile0 + blockIdx.x, t = threadIdx.x;
  int own_slot = 0;
  if (c.use_coarse) {
    if (reduced) {  // partitioned solve: P^T x summed over the ranks already
      if (t < c.ncoarse) pcs[t] = reduced[t];
    } else if (t < c.ncoarse) {  // fixed-order sum over the slots, eight requests in flight at a time
      double sacc = 0.0;
      for (int q0 = 0; q0 < c.tile_slots; q0 += 8) {
        double w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = part[cpart_index(c, min(q0 + u, c.tile_slots - 1), t / NF, t - (t / NF) * NF, NF)];
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc += (q0 + u < c.tile_slots) ? w[u] : 0.0;
      }
      pcs[t] = sacc;
    }
    __syncthreads();
    CoarseRows<NF> crow;
    crow.load(c, tile);
    crow.apply(c, tile, pcs, ycl);
    for (int z = 0; z < c.tile_nagg[tile]; ++z) if (c.tile_aggs[tile * kTileAggs + z] == c.tile_agg[tile]) own_slot = z;
  } else if (t < kTileAggs * NF) {
    ycl[t] = 0.0;
  }
  __syncthreads();
  const int sl = t >> 6, lane = t & 63;
  bool active = false; int I = 0, i = 0, Iloc = 0;
  if (sl < kSlicesPerTile) {
    const int s = c.tile_slice0[tile] + sl;
    Iloc = lane / NF; i = lane - Iloc * NF;
    active = s < c.tile_slice0[tile + 1] && Iloc < c.slice_nn[s];
    if (active) { I = c.slice_node0[s] + Iloc; xv[sl][lane] = x[I * NF + i] + ycl[own_slot * NF + i]; }
  }
  __syncthreads();
  if (!active) return;
  const double* d = c.Dinv + ((size_t)I * NF + i) * NF;
  double z = 0.0;
#pragma unroll
  for (int mI = 0; mI < NF; ++mI) z += d[mI] * xv[sl][Iloc * NF + mI];
  const int r = I * NF + i;
  const double dx = (scale_dst == 0.0 ? 0.0 : scale_dst * dst[r]) + scale_x * z;
  if (upd.u) {
    upd.u[r] -= upd.omega * dx;
    dst[r] = upd.a * dx + (upd.b != 0.0 ? upd.b * upd.xp[r] : 0.0);
    upd.xp[r] = dx;
  } else {
    dst[r] = dx;
  }
}

// bandwidth probe: stream n doubles (16 B per lane) and keep one checksum per workgroup
__global__ __launch_bounds__(256) void k_stream_read(const double2* __restrict__ a, size_t n2, double* __restrict__ out) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const double2 v = a[i]; acc += v.x + v.y; }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0 && acc == 123.456) out[blockIdx.x] = acc;
}

// True residual of a finished Krylov solve: r = b - Ax (Ax from k_spmv_plain), per-workgroup partials of ||r||^2.
__global__ __launch_bounds__(kVecBlock) void k_true_residual(const double* __restrict__ b, const double* __restrict__ ax,
                                                             double* __restrict__ r, double* __restrict__ part, int n) {
  __shared__ double lds[4];
  const int i = blockIdx.x * kVecBlock + threadIdx.x;
  double v = 0.0;
  if (i < n) { v = b[i] - ax[i]; r[i] = v; }
  double w[1] = {v * v};
  block_sum<1>(w, lds);
  if (threadIdx.x == 0) part[blockIdx.x] = w[0];
}

// Test of a warm start: partials of (w,b), (w,w), (b,b) with w = J x0 (from k_spmv_plain), three per workgroup.
__global__ __launch_bounds__(kVecBlock) void k_dots3(const double* __restrict__ w, const double* __restrict__ b,
                                                     double* __restrict__ part, int n, int nblocks, int lo = 0, int hi = 0x7fffffff) {
  __shared__ double lds[12];
  const int i = blockIdx.x * kVecBlock + threadIdx.x;
  double v[3] = {0.0, 0.0, 0.0};
  if (i < n && i >= lo && i < hi) { const double wi = w[i], bi = b[i]; v[0] = wi * bi; v[1] = wi * wi; v[2] = bi * bi; }   // [lo, hi): the owned dofs of a partitioned handle
  block_sum<3>(v, lds);
  if (threadIdx.x == 0) { part[blockIdx.x] = v[0]; part[nblocks + blockIdx.x] = v[1]; part[2 * nblocks + blockIdx.x] = v[2]; }
}
// start of a warm solve: r0 = b - w with w = J x0 (x0 stays in kx)
__global__ void k_start_residual(double* __restrict__ r, const double* __restrict__ b, const double* __restrict__ w, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) r[i] = b[i] - w[i];
}

// deterministic pseudo-random vector in [-1, 1) (shadow vector of a BiCGStab pass that follows a breakdown)
__global__ void k_fill_hash(double* __restrict__ v, unsigned seed, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    v[i] = (double)h * (2.0 / 4294967296.0) - 1.0;
  }
}

// warm start of the next Newton correction: x <- a x + b xp, xp <- old x  (x = dx_k, xp = dx_{k-1})
// u -= omega dx and the predicted start of the next linear solve in one pass (what k_minv_apply does itself at the
// normal end of a solve; this kernel serves the checked / direct-fallback ends)
__global__ void k_update_predict(double* __restrict__ u, double* __restrict__ x, double* __restrict__ xp, double omega, double a, double b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double t = x[i]; u[i] -= omega * t; x[i] = a * t + (b != 0.0 ? b * xp[i] : 0.0); xp[i] = t; }
}
__global__ void k_warm_start(double* __restrict__ x, double* __restrict__ xp, double a, double b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double t = x[i]; x[i] = a * t + (b != 0.0 ? b * xp[i] : 0.0); xp[i] = t; }
}

__global__ void k_axpy(double* __restrict__ y, const double* __restrict__ x, double a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}

// Start of a BiCGStab solve in ONE launch (was: copy, two memsets, k_restrict and a host-to-device copy of the scalars):
// rhat = shadow vector (r_0 unless given), y = 0, restriction partials of r_0 where A(0) expects them, hand-over flags
// cleared, scalars set from the kernel argument.  One workgroup per tile, as k_restrict.
template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_krylov_init(const Ctx c, const double* __restrict__ shadow, const KrylovScalars init,
                                                           double* __restrict__ part) {
  __shared__ double lv[kSlicesPerTile * 64];
  const int tile = c.tile0 + blockIdx.x, t = threadIdx.x;
  if (t < kSlicesPerTile * 64) {
    const int sl = t >> 6, lane = t & 63, s = c.tile_slice0[tile] + sl;
    const int Iloc = lane / NF, i = lane - Iloc * NF;
    const bool active = s < c.tile_slice0[tile + 1] && Iloc < c.slice_nn[s] && Iloc < kWave / NF;
    double v = 0.0;
    if (active) {
      const int idx = (c.slice_node0[s] + Iloc) * NF + i;
      v = c.kr[idx];
      c.krhat[idx] = shadow ? shadow[idx] : v;
      c.ky[idx] = 0.0;
    }
    lv[t] = v;
  }
  if (blockIdx.x == 0) {
    for (int q = t; q < 16 * 66; q += kVecBlock) c.ticket[q] = 0u;
    if (t == 0) *c.scal = init;
  }
  __syncthreads();
  if (c.use_coarse && t < NF) {
    double sacc = 0.0;
    for (int q = 0; q < kSlicesPerTile; ++q)
      for (int il = 0; il < kWave / NF; ++il) sacc += lv[q * 64 + il * NF + t];
    part[cpart_index(c, c.tile_slot[tile], c.tile_agg[tile], t, NF)] = sacc;
  }
}
__global__ void k_copy2(double* __restrict__ a, double* __restrict__ b, const double* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double v = src[i]; a[i] = v; if (b) b[i] = v; }
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

// Solve Dm X = [Lm | Um | bm] for row j of level lv with EIGHT LANES per system: lane r (< NF) of the 8-lane group
// holds row r of the augmented matrix in registers, W[0..NF) = Dm row, W[NF..NC) = right-hand sides.  Gauss-Jordan
// with partial pivoting, rows stay where they are: in step k the lane with the largest |W[k]| among the rows not used yet
// is the pivot row (the search is three shuffle rounds on ONE 64-bit key: the bit pattern of |W[k]| with its last three
// mantissa bits replaced by 7 - lane, so that equal values go to the lower lane), its row is broadcast once per column and
// eliminated from all other rows; lane k notes where variable k ended up and one last shuffle per right-hand side brings the
// solution rows home.  169 shuffles per system instead of the 308 of the row-swapping form (two broadcasts per column): the
// chain of a reduction level is mostly these shuffles (9.8 -> 8.2 us per level on the 50 um mesh).  On return W[NF..NC) of lane r
// is row r of Dm^{-1}[Lm | Um | bm].  Every lane of the wave must call it (inactive groups pass on = false).
template <int NF>
__device__ inline bool tri_group_solve(const TriLevel& lv, int j, bool on, int r, double (&W)[3 * NF + 1]) {
  constexpr int NC = 3 * NF + 1;
  static_assert(NF <= 8, "one lane per row of an 8-lane group");
  const bool rowon = on && r < NF;
  const int jc = on ? j : 0, rc = r < NF ? r : 0;
#pragma unroll
  for (int cI = 0; cI < NF; ++cI) {
    const size_t off = (size_t)(rc * NF + cI) * lv.n + jc;
    W[cI] = lv.D[off]; W[NF + cI] = lv.L[off]; W[2 * NF + cI] = lv.U[off];
  }
  W[3 * NF] = lv.b[(size_t)rc * lv.n + jc];
  if (!rowon) {  // identity row: never chosen as a pivot for another column, harmless in the shuffles
#pragma unroll
    for (int cI = 0; cI < NC; ++cI) W[cI] = (cI == r && r < NF) ? 1.0 : 0.0;
  }
  bool bad = false, used = r >= NF;
  int home = r;   // lane that ends up holding the solution row of variable r
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    unsigned long long key = used ? 0ull : (((unsigned long long)__double_as_longlong(fabs(W[k])) & ~7ull) | (unsigned long long)(7 - r));
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      const int olo = __shfl_xor((int)(unsigned)key, o, 8), ohi = __shfl_xor((int)(unsigned)(key >> 32), o, 8);
      const unsigned long long ok_ = ((unsigned long long)(unsigned)ohi << 32) | (unsigned)olo;
      key = ok_ > key ? ok_ : key;
    }
    const int p = 7 - (int)(key & 7ull);
    bad |= (key >> 3) == 0ull;   // the largest candidate is (all but) zero
    const double ip = 1.0 / __shfl(W[k], p, 8);
    const double f = (r == p) ? 0.0 : W[k] * ip;
#pragma unroll
    for (int cI = k + 1; cI < NC; ++cI) {
      const double from_p = __shfl(W[cI], p, 8);
      W[cI] = (r == p) ? W[cI] * ip : W[cI] - f * from_p;
    }
    W[k] = (r == p) ? 1.0 : 0.0;
    used = used || (r == p);
    home = (r == k) ? p : home;
  }
#pragma unroll
  for (int cI = NF; cI < NC; ++cI) W[cI] = __shfl(W[cI], home, 8);
  return !(on && bad);
}

// One level of the reduction.  16 lanes per row ih of the upper level: lanes 0-7 eliminate the left neighbour
// (row 2ih-1 of the lower level), lanes 8-15 the right one (2ih+1); lane r of each half owns row r of the 7x7 blocks.
template <int NF>
__device__ __forceinline__ void bcr_forward_row(const TriLevel& lo, const TriLevel& hi, int32_t* status, const int ih_raw) {
  constexpr int NC = 3 * NF + 1;
  const int t = threadIdx.x;
  const int side = (t >> 3) & 1, r = t & 7;
  const bool rowok = ih_raw < hi.n;
  const int ih = rowok ? ih_raw : 0, i = 2 * ih;
  const int j = side == 0 ? i - 1 : i + 1;
  const bool on = rowok && j >= 0 && j < lo.n;
  const int rc = r < NF ? r : 0;
  double W[NC];
  // coupling of row i to row j: row r of lo.L (left) / lo.U (right); own row of lo.D and lo.b
  double cr[NF], dr[NF];
  const double* C = side == 0 ? lo.L : lo.U;
#pragma unroll
  for (int mI = 0; mI < NF; ++mI) { cr[mI] = C[(size_t)(rc * NF + mI) * lo.n + i]; dr[mI] = lo.D[(size_t)(rc * NF + mI) * lo.n + i]; }
  const double br = lo.b[(size_t)rc * lo.n + i];
  const bool ok = tri_group_solve<NF>(lo, j, on, r, W);
  // products C * X: X row m sits in lane m of this half
  double pL[NF], pU[NF], pb = 0.0;
#pragma unroll
  for (int cI = 0; cI < NF; ++cI) { pL[cI] = 0.0; pU[cI] = 0.0; }
#pragma unroll
  for (int mI = 0; mI < NF; ++mI) {
    const double cm = on ? cr[mI] : 0.0;
#pragma unroll
    for (int cI = 0; cI < NF; ++cI) { pL[cI] += cm * __shfl(W[NF + cI], mI, 8); pU[cI] += cm * __shfl(W[2 * NF + cI], mI, 8); }
    pb += cm * __shfl(W[3 * NF], mI, 8);
  }
  // the diagonal block and the right-hand side take a term from both halves
  double dmine[NF], dother[NF];
#pragma unroll
  for (int cI = 0; cI < NF; ++cI) { dmine[cI] = side == 0 ? pU[cI] : pL[cI]; dother[cI] = __shfl_xor(dmine[cI], 8, 16); }
  const double pbo = __shfl_xor(pb, 8, 16);
  if (rowok && r < NF) {
    if (side == 0) {
#pragma unroll
      for (int cI = 0; cI < NF; ++cI) {
        hi.L[(size_t)(r * NF + cI) * hi.n + ih] = -pL[cI];
        hi.D[(size_t)(r * NF + cI) * hi.n + ih] = (dr[cI] - dmine[cI]) - dother[cI];  // left term first, then right
      }
    } else {
#pragma unroll
      for (int cI = 0; cI < NF; ++cI) hi.U[(size_t)(r * NF + cI) * hi.n + ih] = -pU[cI];
      hi.b[(size_t)r * hi.n + ih] = (br - pbo) - pb;
      if (on) {  // keep the right neighbour's solved couplings for the way back up
#pragma unroll
        for (int cI = 0; cI < NF; ++cI) {
          lo.Li[(size_t)(r * NF + cI) * lo.n + j] = W[NF + cI];
          lo.Ui[(size_t)(r * NF + cI) * lo.n + j] = W[2 * NF + cI];
        }
        lo.bi[(size_t)r * lo.n + j] = W[3 * NF];
      }
    }
  }
  if (!ok) atomicOr(status, 2);
}
template <int NF>
__global__ __launch_bounds__(64) void k_bcr_forward(TriLevel lo, TriLevel hi, int32_t* status) {
  bcr_forward_row<NF>(lo, hi, status, blockIdx.x * 4 + ((int)threadIdx.x >> 4));
}

// top of the pyramid: one row, x = D^{-1} b
template <int NF>
__device__ __forceinline__ void bcr_top_row(const TriLevel& top, int32_t* status) {
  constexpr int NC = 3 * NF + 1;
  const int t = threadIdx.x, r = t & 7;
  double W[NC];
  const bool ok = tri_group_solve<NF>(top, 0, t < 8, r, W);
  if (t < NF) top.x[t] = W[3 * NF];
  if (!ok) atomicOr(status, 2);
}
template <int NF>
__global__ __launch_bounds__(64) void k_bcr_top(TriLevel top, int32_t* status) { bcr_top_row<NF>(top, status); }

// One level of the way back: x of the even rows is the upper level's, an odd row j gets x_j = bi_j - Li_j x_{j-1} - Ui_j x_{j+1}.
// One THREAD per (row, field): its 2 NF + 1 loads are one round trip (one thread per row ran NF of them one after the other:
// 8 us per level instead of 3).
template <int NF>
__device__ __forceinline__ void bcr_backward_entry(const TriLevel& lo, const TriLevel& hi, const int q) {
  if (q >= lo.n * NF) return;
  const int r = q / lo.n, j = q - r * lo.n;   // j fastest: the loads of a wave are contiguous
  if ((j & 1) == 0) { lo.x[(size_t)r * lo.n + j] = hi.x[(size_t)r * hi.n + (j >> 1)]; return; }
  const bool has_r = (j + 1 < lo.n);
  double li[NF], ui[NF], xl[NF], xr[NF];
#pragma unroll
  for (int cI = 0; cI < NF; ++cI) {
    li[cI] = lo.Li[(size_t)(r * NF + cI) * lo.n + j]; ui[cI] = lo.Ui[(size_t)(r * NF + cI) * lo.n + j];
    xl[cI] = hi.x[(size_t)cI * hi.n + ((j - 1) >> 1)];
    xr[cI] = has_r ? hi.x[(size_t)cI * hi.n + ((j + 1) >> 1)] : 0.0;
  }
  double sacc = lo.bi[(size_t)r * lo.n + j];
#pragma unroll
  for (int cI = 0; cI < NF; ++cI) sacc -= li[cI] * xl[cI] + ui[cI] * xr[cI];
  lo.x[(size_t)r * lo.n + j] = sacc;
}
template <int NF>
__global__ __launch_bounds__(kVecBlock) void k_bcr_backward(TriLevel lo, TriLevel hi) {
  bcr_backward_entry<NF>(lo, hi, blockIdx.x * kVecBlock + threadIdx.x);
}

// The TOP of the pyramid in one launch: the levels whose upper neighbour has at most kBcrTailRows rows, the single-row solve, and the
// same levels on the way back — seven launches become one WAVE that walks through them with a barrier in between (a level's
// results are read by the same wave: its own CU's cache serves them).  One wave and no more: the forward step of a level is a chain
// of shuffles through the CU's LDS crossbar, and the waves of ONE workgroup share one crossbar where the waves of separate launches
// have a CU each — measured on the 50 um mesh (whole solve): per-level launches 160.6 us, tail of the levels up to 4 rows in one wave
// 156.2, up to 16 rows in four waves 168, up to 47 rows in eight waves 225.  The arithmetic is that of the per-level kernels (same
// device functions).
constexpr int kBcrTailRows = 4, kBcrTailLevels = 8, kBcrTailThreads = 64;   // ONE wave: its shuffles have the CU's LDS crossbar to themselves
struct TriTail { TriLevel lv[kBcrTailLevels + 1]; int nlev; };   // lv[0] = the lowest level of the tail ... lv[nlev - 1] = the single row
template <int NF>
__global__ __launch_bounds__(kBcrTailThreads) void k_bcr_tail(const TriTail tt, int32_t* status) {
  const int t = threadIdx.x;
  for (int l = 0; l + 1 < tt.nlev; ++l) {
    for (int ih0 = 0; ih0 < tt.lv[l + 1].n; ih0 += kBcrTailThreads / 16)   // (uniform trip count; rows beyond the level's end are masked inside)
      bcr_forward_row<NF>(tt.lv[l], tt.lv[l + 1], status, ih0 + (t >> 4));
    __syncthreads();
  }
  if (t < 64) bcr_top_row<NF>(tt.lv[tt.nlev - 1], status);
  __syncthreads();
  for (int l = tt.nlev - 2; l >= 0; --l) {
    for (int q = t; q < tt.lv[l].n * NF; q += kBcrTailThreads) bcr_backward_entry<NF>(tt.lv[l], tt.lv[l + 1], q);
    __syncthreads();
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
  const int s = c.node_slice[I], il = I - c.slice_node0[s];
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
