// Consistent-mass L2 projection onto P1 on the device: the reference's post-processing
//     project(-grad(u_np), W)                     1D/MPNP_CO2ER_EDL.py:802-805
//     project(grad(u_nX), W)  for all nine fields  3D/MPNP_CO2ER_pore.py:884-909
//     project(CellDiameter(mesh)), project(sqrt(inner(grad(u_np), grad(u_np))))   1D:599,651-653 (SUPG parameters)
// [3P] DOLFIN's project() assembles the P1 mass matrix and solves M g = int f phi; here: the mass matrix on the node pattern
// of the Jacobian (one double per node block, built once per handle from the per-block contribution lists: no atomics), the
// right-hand side by the node -> element gather, and Jacobi-preconditioned CG for all components at once (the diagonally
// scaled P1 mass matrix has a condition number <= d + 2 whatever the mesh grading, so 1e-14 takes about 30 iterations).
// Included at the end of gmpnp_api.hip.
#pragma once

namespace gmpnp {

constexpr int kProjMaxComp = 4;
struct ProjScal { double v[kProjMaxComp]; };

template <int DIM>
__device__ inline void proj_cell_geometry(const Ctx& c, int e, double& vol, double (&g)[DIM + 1][DIM]) {
  constexpr int NN = DIM + 1;
  double X[NN][DIM];
#pragma unroll
  for (int a = 0; a < NN; ++a) {
    const int nd = c.cells[e * NN + a];
#pragma unroll
    for (int d = 0; d < DIM; ++d) X[a][d] = c.coords[(size_t)nd * DIM + d];
  }
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
}

// cell volumes and, with f, the cell-wise constant sign * grad f
template <int DIM>
__global__ __launch_bounds__(256) void k_proj_cells(const Ctx c, const double* __restrict__ f, double sign, double* __restrict__ cellvol,
                                                     double* __restrict__ cellval) {
  constexpr int NN = DIM + 1;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= c.nc) return;
  double vol, g[NN][DIM];
  proj_cell_geometry<DIM>(c, e, vol, g);
  cellvol[e] = vol;
  if (f) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < NN; ++a) s += f[c.cells[e * NN + a]] * g[a][d];
      cellval[(size_t)e * DIM + d] = sign * s;
    }
  }
}

// mass[k] = sum over the element contributions of node block k of |K| (1 + delta_ab) / ((d+1)(d+2)); diag[I] = M_II
template <int DIM>
__global__ __launch_bounds__(256) void k_proj_mass(const Ctx c, const double* __restrict__ cellvol, double* __restrict__ mass, double* __restrict__ diag) {
  const int I = blockIdx.x * 256 + threadIdx.x;
  if (I >= c.nv) return;
  constexpr double MDEN = 1.0 / ((DIM + 1) * (DIM + 2));
  for (int k = c.rowptr[I]; k < c.rowptr[I + 1]; ++k) {
    double s = 0.0;
    for (int q = c.cptr[k]; q < c.cptr[k + 1]; ++q) {
      const int pk = c.contrib[q];
      s += cellvol[pk >> 4] * MDEN * (((pk >> 2) & 3) == (pk & 3) ? 2.0 : 1.0);
    }
    mass[k] = s;
    if (c.cols[k] == I) diag[I] = s;
  }
}

// b[I][cmp] = sum over the cells around node I of |K| / (d+1) * cellval[e][cmp]
template <int DIM>
__global__ __launch_bounds__(256) void k_proj_rhs(const Ctx c, const double* __restrict__ cellvol, const double* __restrict__ cellval, int ncomp,
                                                   double* __restrict__ b) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= c.nv * ncomp) return;
  const int I = t / ncomp, cmp = t - I * ncomp;
  double s = 0.0;
  for (int k = c.n2e_ptr[I]; k < c.n2e_ptr[I + 1]; ++k) {
    const int e = c.n2e[k] / (DIM + 1);
    s += cellvol[e] * (1.0 / (DIM + 1)) * cellval[(size_t)e * ncomp + cmp];
  }
  b[t] = s;
}

__global__ __launch_bounds__(256) void k_proj_spmv(const Ctx c, const double* __restrict__ mass, const double* __restrict__ x, int ncomp, double* __restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= c.nv * ncomp) return;
  const int I = t / ncomp, cmp = t - I * ncomp;
  double s = 0.0;
  for (int k = c.rowptr[I]; k < c.rowptr[I + 1]; ++k) s += mass[k] * x[(size_t)c.cols[k] * ncomp + cmp];
  y[t] = s;
}

// per-workgroup partial sums of (a, b) for every component: part[cmp * nblocks + block]
__global__ __launch_bounds__(256) void k_proj_dots(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ scale,
                                                    int nv, int ncomp, double* __restrict__ part, int nblocks) {
  __shared__ double lds[4 * kProjMaxComp];
  const int I = blockIdx.x * 256 + threadIdx.x;
  double v[kProjMaxComp];
#pragma unroll
  for (int q = 0; q < kProjMaxComp; ++q) v[q] = 0.0;
  if (I < nv) {
    const double w = scale ? 1.0 / scale[I] : 1.0;   // scale = diagonal of M: (r, D^-1 r)
    for (int q = 0; q < ncomp; ++q) v[q] = a[(size_t)I * ncomp + q] * b[(size_t)I * ncomp + q] * w;
  }
  block_sum<kProjMaxComp>(v, lds);
  if (threadIdx.x == 0)
    for (int q = 0; q < ncomp; ++q) part[(size_t)q * nblocks + blockIdx.x] = v[q];
}

// x += alpha p ; r -= alpha Ap
__global__ __launch_bounds__(256) void k_proj_update(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p, const double* __restrict__ Ap,
                                                      const ProjScal alpha, int nv, int ncomp) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nv * ncomp) return;
  const double al = alpha.v[t % ncomp];
  x[t] += al * p[t]; r[t] -= al * Ap[t];
}
// p = D^-1 r + beta p
__global__ __launch_bounds__(256) void k_proj_direction(double* __restrict__ p, const double* __restrict__ r, const double* __restrict__ diag,
                                                         const ProjScal beta, int nv, int ncomp) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nv * ncomp) return;
  p[t] = r[t] / diag[t / ncomp] + beta.v[t % ncomp] * p[t];
}

}  // namespace gmpnp

namespace {

template <int DIM>
int project_on_device(gmpnp_solver* s, gmpnp_projector& P, int ncomp, const double* nodal_file_order, double sign,
                      const double* cell_values, double* out, gmpnp_linear_stats_t* st) {
  using namespace gmpnp;
  const int nv = s->t.nv, nc = s->t.nc, n = nv * ncomp;
  if (!P.h_part) {
    P.nblocks = grid_for(nv, 256);
    HIP_TRY(hipHostMalloc((void**)&P.h_part, (size_t)kProjMaxComp * P.nblocks * sizeof(double)));
    HIP_TRY(P.cellvol.alloc(nc)); HIP_TRY(P.cellval.alloc((size_t)nc * kProjMaxComp)); HIP_TRY(P.mass.alloc(s->nb)); HIP_TRY(P.diag.alloc(nv));
    HIP_TRY(P.f.alloc(nv));
    for (DevBuf<double>* v : {&P.b, &P.x, &P.r, &P.p, &P.Ap}) HIP_TRY(v->alloc((size_t)nv * kProjMaxComp));
    HIP_TRY(P.dpart.alloc((size_t)kProjMaxComp * P.nblocks));
  }
  hipStream_t q = s->stream;
  HIP_TRY(hipStreamSynchronize(q));
  std::vector<double> host((size_t)std::max(nv, nc * ncomp));
  if (nodal_file_order) {
    for (int I = 0; I < nv; ++I) host[I] = nodal_file_order[s->t.perm[I]];
    HIP_TRY(hipMemcpy(P.f.p, host.data(), nv * sizeof(double), hipMemcpyHostToDevice));
  } else {
    HIP_TRY(hipMemcpy(P.cellval.p, cell_values, (size_t)nc * ncomp * sizeof(double), hipMemcpyHostToDevice));   // cell order = file order
  }
  hipLaunchKernelGGL((k_proj_cells<DIM>), dim3(grid_for(nc, 256)), dim3(256), 0, q, s->c, nodal_file_order ? (const double*)P.f.p : (const double*)nullptr,
                     sign, P.cellvol.p, P.cellval.p);
  if (!P.mass_ready) {
    hipLaunchKernelGGL((k_proj_mass<DIM>), dim3(grid_for(nv, 256)), dim3(256), 0, q, s->c, (const double*)P.cellvol.p, P.mass.p, P.diag.p);
    P.mass_ready = true;
  }
  const dim3 gv(grid_for(n, 256)), gn(P.nblocks);
  hipLaunchKernelGGL((k_proj_rhs<DIM>), gv, dim3(256), 0, q, s->c, (const double*)P.cellvol.p, (const double*)P.cellval.p, ncomp, P.b.p);
  HIP_TRY(hipMemsetAsync(P.x.p, 0, (size_t)n * sizeof(double), q));
  HIP_TRY(hipMemcpyAsync(P.r.p, P.b.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, q));
  HIP_TRY(hipMemsetAsync(P.p.p, 0, (size_t)n * sizeof(double), q));
  auto dot = [&](const double* a, const double* b2, const double* scale, double* res) -> int {
    hipLaunchKernelGGL(k_proj_dots, gn, dim3(256), 0, q, a, b2, scale, nv, ncomp, P.dpart.p, P.nblocks);
    HIP_TRY(hipMemcpyAsync(P.h_part, P.dpart.p, (size_t)ncomp * P.nblocks * sizeof(double), hipMemcpyDeviceToHost, q));
    HIP_TRY(hipStreamSynchronize(q));
    for (int cmp = 0; cmp < ncomp; ++cmp) {
      double acc = 0.0;
      for (int i = 0; i < P.nblocks; ++i) acc += P.h_part[(size_t)cmp * P.nblocks + i];
      res[cmp] = acc;
    }
    return GMPNP_OK;
  };
  double bb[kProjMaxComp], rz[kProjMaxComp], rr[kProjMaxComp], pAp[kProjMaxComp];
  int rc = dot(P.b.p, P.b.p, nullptr, bb); if (rc) return rc;
  rc = dot(P.r.p, P.r.p, P.diag.p, rz); if (rc) return rc;
  ProjScal beta{}, alpha{};
  int it = 0;
  double worst = 0.0;
  const int maxit = 500;
  for (; it < maxit; ++it) {
    hipLaunchKernelGGL(k_proj_direction, gv, dim3(256), 0, q, P.p.p, (const double*)P.r.p, (const double*)P.diag.p, beta, nv, ncomp);
    hipLaunchKernelGGL(k_proj_spmv, gv, dim3(256), 0, q, s->c, (const double*)P.mass.p, (const double*)P.p.p, ncomp, P.Ap.p);
    rc = dot(P.p.p, P.Ap.p, nullptr, pAp); if (rc) return rc;
    for (int cmp = 0; cmp < ncomp; ++cmp) alpha.v[cmp] = (pAp[cmp] > 0.0 && rz[cmp] > 0.0) ? rz[cmp] / pAp[cmp] : 0.0;
    hipLaunchKernelGGL(k_proj_update, gv, dim3(256), 0, q, P.x.p, P.r.p, (const double*)P.p.p, (const double*)P.Ap.p, alpha, nv, ncomp);
    double rz_new[kProjMaxComp];
    rc = dot(P.r.p, P.r.p, P.diag.p, rz_new); if (rc) return rc;
    rc = dot(P.r.p, P.r.p, nullptr, rr); if (rc) return rc;
    worst = 0.0;
    for (int cmp = 0; cmp < ncomp; ++cmp) {
      beta.v[cmp] = rz[cmp] > 0.0 ? rz_new[cmp] / rz[cmp] : 0.0;
      rz[cmp] = rz_new[cmp];
      if (bb[cmp] > 0.0) worst = std::max(worst, std::sqrt(rr[cmp] / bb[cmp]));
    }
    if (!(worst > 1e-14)) { ++it; break; }
  }
  if (st) { st->iterations = it; st->converged = worst <= 1e-12 ? 1 : 0; st->residual_norm = worst; st->rhs_norm = 1.0; }
  if (!(worst <= 1e-12)) return fail(GMPNP_ERR_LINEAR, "mass-matrix CG of the projection did not reach 1e-12");
  HIP_TRY(hipMemcpy(host.data(), P.x.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  for (int I = 0; I < nv; ++I)
    for (int cmp = 0; cmp < ncomp; ++cmp) out[(size_t)s->t.perm[I] * ncomp + cmp] = host[(size_t)I * ncomp + cmp];
  return GMPNP_OK;
}

}  // namespace

extern "C" {

int gmpnp_project_gradient(gmpnp_solver* s, const double* nodal_values, double sign, double* out, gmpnp_linear_stats_t* stats) {
  if (!s || !nodal_values || !out) return fail(GMPNP_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  if (!s->projector) s->projector.reset(new gmpnp_projector);
  return s->dim == 3 ? project_on_device<3>(s, *s->projector, 3, nodal_values, sign, nullptr, out, stats)
                     : project_on_device<1>(s, *s->projector, 1, nodal_values, sign, nullptr, out, stats);
}

int gmpnp_project_cellwise(gmpnp_solver* s, int32_t ncomp, const double* cell_values, double* out, gmpnp_linear_stats_t* stats) {
  if (!s || !cell_values || !out) return fail(GMPNP_ERR_INVALID, "NULL argument");
  if (ncomp < 1 || ncomp > gmpnp::kProjMaxComp) return fail(GMPNP_ERR_INVALID, "ncomp must be 1..4");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  if (!s->projector) s->projector.reset(new gmpnp_projector);
  return s->dim == 3 ? project_on_device<3>(s, *s->projector, ncomp, nullptr, 1.0, cell_values, out, stats)
                     : project_on_device<1>(s, *s->projector, ncomp, nullptr, 1.0, cell_values, out, stats);
}

}  // extern "C"
