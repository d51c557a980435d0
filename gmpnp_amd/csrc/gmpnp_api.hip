// C ABI of libgmpnp.so (include/gmpnp.h): handle, host drivers of assembly, preconditioner setup,
// BiCGStab and the damped Newton loop ([3P] dolfin::NewtonSolver semantics).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include <hip/hip_ext.h>

#include "gmpnp_kernels.h"
#include "gmpnp_band_lu.h"
#include "gmpnp_dist_kernels.h"
#include "gmpnp_multilevel.h"

using namespace gmpnp;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return fail(GMPNP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));              \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t count, bool zero = true, size_t pad = 0) {  // pad: zeroed elements behind the n counted ones
    if (p) { (void)hipFree(p); p = nullptr; }
    n = count;
    hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count + pad, 1) * sizeof(T));
    if (e == hipSuccess && (zero || pad)) e = hipMemset(p, 0, std::max<size_t>(count + pad, 1) * sizeof(T));
    return e;
  }
  hipError_t upload(const std::vector<T>& v) {
    hipError_t e = alloc(v.size(), v.empty());
    if (e == hipSuccess && !v.empty()) e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
  }
};

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// device side of gmpnp_project_gradient / gmpnp_project_cellwise (gmpnp_project.h), allocated on first use
struct gmpnp_projector {
  DevBuf<double> cellvol, cellval, mass, diag, f, b, x, r, p, Ap, dpart;
  double* h_part = nullptr;   // pinned: dot-product partials
  int nblocks = 0;
  bool mass_ready = false;
  ~gmpnp_projector() { if (h_part) (void)hipHostFree(h_part); }
};

struct gmpnp_solver {
  Topology t;
  gmpnp_model_t model{};
  gmpnp_quadrature_t quad{};
  gmpnp_options_t opts{};
  int dim = 0, nf = 0, nn = 0, ndof = 0, nb = 0, ncoarse = 0;
  int n_resblocks = 0;
  hipStream_t stream = nullptr;
  Ctx c{};
  // boundary facets kept on the host (internal vertex ids) to rebuild flux tables when the model changes
  std::vector<int32_t> wall_f, exit_f, point_v;
  // device storage
  DevBuf<gmpnp_model_t> d_model; DevBuf<gmpnp_quadrature_t> d_quad;
  DevBuf<double> coords, u, un, F, bcval, bndF, rob_val, EF, EJ, vals, vals_s, Dinv, AP, AcPart, Ac, Aci;
  DevBuf<double> kr, krhat, kp0, kp1, kv0, kv1, ks, kt, ky, kx, kxp, kb, yc, cpart_r0, cpart_r1, cpart_p0, cpart_p1, cpart_v0, cpart_v1, cpart_t,
      part_a, part_b, part_f;
  DevBuf<int32_t> cells, robF_ptr, rob_col, rob_row, n2e_ptr, n2e, rowptr, cols, cptr, contrib, slice_colbase,
      slice_node0, slice_nn, node_slice, sell_cols, sell_blk, wl_slice, wl_kpos, tile_slice0, tile_agg, tile_slot,
      tile_aggs, tile_nagg, tile_cols, tile_colslot, sell_lcol, agg, agg_start, row_aggs, status;
  DevBuf<int64_t> rob_addr, slice_off;
  DevBuf<uint8_t> bcflag, sell_aggslot;
  DevBuf<KrylovScalars> scal; DevBuf<TileRec> tile_rec;
  std::vector<uint8_t> h_bcflag, h_bcflag_dev;   // flags being built / flags the device holds
  double* h_bcval = nullptr;    // pinned [ndof]
  double* h_stage = nullptr;    // pinned [ndof]: staging of the file-order <-> internal-order vector transfers
  // pinned read-back areas
  KrylovScalars* h_scal = nullptr; double* h_part = nullptr; int32_t* h_status = nullptr;
  HostPoll* h_poll = nullptr;   // progress mirror the B kernels write (fine-grained pinned memory)
  int host_poll = 1;            // opts.progress_by_copy: poll with a device-to-host copy + event per burst instead
  int burst_iters = 1;  // iterations per polling burst (opts.burst_iterations); with copy + event polling: 1 -> 453, 2 -> 463,
                        // 4 -> 456, 8 -> 436 Newton its/s; with the pinned progress mirror a poll costs nothing on the
                        // device: 1 -> 496, 2 -> 492
  int krylov_hint = 0;  // expected iterations of the next solve (the same Newton iteration of the previous time step), 0 = none
  int hint_by_newton_it[32] = {0};
  int last_krylov_iters[2] = {0, 0};
  bool jacobian_valid = false, precond_valid = false;
  int precond_mode = -1;
  const double* shadow_src = nullptr; double shadow_rho0 = 0.0;  // shadow vector of the next krylov() pass (after a breakdown: krand)
  int last_done = 0;            // exit code of the last device Krylov loop (1 converged, 2 iteration cap, 3 breakdown / divergence)
  DevBuf<double> krand;         // pseudo-random shadow vector for a pass that follows a breakdown
  bool state_jumped = true;     // u was set from outside since the last Newton solve: the Jacobian moves a lot, no coarse reuse
  bool coarse_refresh_due = false;  // a solve with a reused coarse inverse took clearly longer than the last fresh one
  int krylov_fresh_iters = 0;   // iterations of the last solve right after a coarse rebuild
  DevBuf<uint32_t> ticket;
  DevBuf<double> supg_rho;  // [nv][ns] internal order
  bool fused_half = false;  // two launches per BiCGStab iteration (coarse workgroups inside the tile launch); opts.launch_form
  int resident_slots = 0;   // workgroups of k_half_a/b the device holds at once (occupancy query at create)
  unsigned fused_seq = 0;   // fused launches so far in the current solve
  int warm_start = 2;  // start Newton iteration k+1's linear solve from (1 - omega) dx_k (+ second-order term); opts.warm_start
  int coarse_lag = 3;   // rebuild the coarse inverse alone every coarse_lag-th Newton iteration of a solve (measured best: 1 -> 3 costs 0.7 % more Krylov iterations and saves 155 us per skipped rebuild)
  // SpMV event sampling (eager mode)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool; size_t ev_used = 0;
  std::vector<int> ev_halves;   // half-iterations inside bracket i (0: the bracket is not counted)
  int64_t solve_index = 0;
  int64_t spmv_launched = 0, spmv_sampled = 0; double spmv_us_sum = 0.0;
  // block-tridiagonal direct solver (1D): cyclic-reduction pyramid
  std::vector<TriLevel> tri; DevBuf<double> tri_store; DevBuf<int32_t> tri_kpos; bool tri_ok = false;
  // block-banded LU (3D): direct solver / fallback of the Krylov solve; storage is allocated on first use
  DevBuf<double> lu_band, lu_dinv, lu_y, lu_part; DevBuf<int32_t> lu_pos, lu_node; BandLU lu{}; bool lu_ready = false;
  // asynchronous coarse refresh: the Galerkin product + inverse of THIS iteration's matrix run on a side stream while
  // BiCGStab uses the inverse built from the previous iteration's matrix; adopted at the next set-up (double buffer)
  hipStream_t stream2 = nullptr; hipEvent_t ev_mat = nullptr, ev_chain = nullptr, ev_jac = nullptr, ev_dots = nullptr;
  int warm_async = 1;         // opts.warm_in_stream: test of the predicted start in the main stream, behind the set-up
  DevBuf<double> Aci2; double* aci_buf[2] = {nullptr, nullptr}; int aci_cur = 0;
  bool chain_in_flight = false;
  int coarse_async = 1;       // opts.coarse_refresh = N: rebuild in the main stream every Nth iteration (the older scheme)
  bool x0_predicted = false;  // kx holds the predicted start of the next linear solve (left by the previous Newton update)
  bool phase_timing = false;  // opts.phase_timing fills ms_assemble / ms_setup / ms_krylov of the Newton statistics
  int direct_fallback = 1;      // opts.no_direct_fallback: a failed Krylov solve is an error again
  int strict_steric = 0;        // opts.strict_steric: 1 - S <= 0 at a quadrature point is fatal (the reference has no such test)
  double lu_max_gb = 48.0;      // opts.band_lu_max_gb: largest band storage the fallback may allocate
  int direct_solves = 0;        // band LU solves since create (factorisations)
  int direct_sticky = 0;        // Newton solves that still go straight to the band LU after a Krylov failure
  int direct_backoff = 0;       // length of the last such stretch (doubles with every new failure, resets on a converged Krylov solve)
  hipEvent_t ev_phase[6] = {};
  hipEvent_t ev_poll[2] = {};
  // mesh partition (gmpnp_create_partition): halo plan in INTERNAL node ids, buffers of the fused exchanges
  bool partitioned = false; int part_rank = 0, part_size = 1;
  bool staged_element = false;   // k_element writes its records through LDS, record by record (large 3D meshes)
  bool matp = false;        // materialised vector form of the BiCGStab half-iterations (opts.vector_form; automatic above 768 MB of matrix)
  bool prereduce = false;   // unpartitioned, many tile slots per aggregate: k_dist_reduce feeds the coarse kernels (Ctx::dist)
  std::vector<int32_t> nb_rank, send_ptr, recv_ptr;   // neighbours; [n_neighbours + 1] offsets into the node lists
  DevBuf<int32_t> send_nodes, recv_nodes, tile_cols_x;
  DevBuf<double> sendbuf, recvbuf, red_i, red_a, red_b, red_norm;
  double* h_red = nullptr;   // pinned [8]: all-reduced ||b||^2 and status bits
  std::unique_ptr<gmpnp_projector> projector;
  // geometric multilevel term (gmpnp_attach_coarse_level, gmpnp_multilevel.h): the link to the next-coarser level (tables in the
  // internal orders of both handles) and this handle's buffers when it serves as a coarse level itself
  gmpnp_solver* ml_coarse = nullptr; double ml_theta = 1.0; bool ml_is_coarse = false;
  int ml_sweeps = 2;   // as the COARSEST level: smoothing steps per application (an intermediate level runs a V(1,1) cycle)
  double ml_omega = 0.7;   // damping of the smoothing steps
  int ml_mid_jacobi = 1;   // intermediate levels smooth with node-block Jacobi alone (the slab coarse space stays with the coarsest level)
  DevBuf<int32_t> ml_par, ml_child_ptr, ml_child, ml_copy;
  DevBuf<double> ml_r, ml_w, ml_z;

  ~gmpnp_solver() {
    for (auto& e : ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& e : ev_phase) if (e) (void)hipEventDestroy(e);
    for (auto& e : ev_poll) if (e) (void)hipEventDestroy(e);
    if (h_scal) (void)hipHostFree(h_scal);
    if (h_poll) (void)hipHostFree(h_poll);
    if (h_bcval) (void)hipHostFree(h_bcval);
    if (h_stage) (void)hipHostFree(h_stage);
    if (h_part) (void)hipHostFree(h_part);
    if (h_status) (void)hipHostFree(h_status);
    if (h_red) (void)hipHostFree(h_red);
    if (stream2) { (void)hipStreamSynchronize(stream2); (void)hipStreamDestroy(stream2); }
    if (ev_mat) (void)hipEventDestroy(ev_mat);
    if (ev_chain) (void)hipEventDestroy(ev_chain);
    if (ev_jac) (void)hipEventDestroy(ev_jac);
    if (ev_dots) (void)hipEventDestroy(ev_dots);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

// ---- dispatch on (dim, n_fields) ---------------------------------------------------------------
#define GMPNP_DISPATCH(s, CALL)                                   \
  do {                                                            \
    if ((s)->dim == 3 && (s)->nf == 9) { constexpr int DIM = 3, NF = 9; CALL; } \
    else if ((s)->dim == 1 && (s)->nf == 7) { constexpr int DIM = 1, NF = 7; CALL; } \
    else return fail(GMPNP_ERR_INVALID, "unsupported (dim, n_fields)"); \
  } while (0)

int grid_for(int n, int block) { return (n + block - 1) / block; }

size_t coarse_lds_bytes(int n, int nf) { return (size_t)(n * n + n * nf + 2 * nf * nf) * sizeof(double); }

// Facet / point integrals that do not depend on u, and the Robin mass entries (SURVEY App. D "facets").
int rebuild_boundary(gmpnp_solver* s) {
  const Topology& t = s->t;
  const int nf = s->nf, ns = nf - 1;
  std::vector<double> bnd((size_t)s->ndof, 0.0);
  struct Ent { int row, col; double v; };
  std::vector<Ent> ents;
  auto area = [&](const int32_t* f) {
    const double* a = &t.coords[(size_t)f[0] * 3]; const double* b = &t.coords[(size_t)f[1] * 3];
    const double* c = &t.coords[(size_t)f[2] * 3];
    const double ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2];
    const double vx = c[0] - a[0], vy = c[1] - a[1], vz = c[2] - a[2];
    const double cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
    return 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
  };
  if (s->dim == 3) {
    for (size_t k = 0; k + 2 < s->wall_f.size(); k += 3) {
      const int32_t* f = &s->wall_f[k]; const double ar = area(f);
      for (int i = 0; i < ns; ++i)
        if (s->model.wall_flux[i] != 0.0)
          for (int a = 0; a < 3; ++a) bnd[(size_t)f[a] * nf + i] += s->model.wall_flux[i] * ar / 3.0;
    }
    for (size_t k = 0; k + 2 < s->exit_f.size(); k += 3) {
      const int32_t* f = &s->exit_f[k]; const double ar = area(f);
      for (int i = 0; i < ns; ++i) {
        const double kap = s->model.exit_kappa[i];
        if (kap == 0.0) continue;
        for (int a = 0; a < 3; ++a) {
          bnd[(size_t)f[a] * nf + i] += -kap * ar / 3.0;
          for (int b = 0; b < 3; ++b) ents.push_back({f[a] * nf + i, f[b] * nf + i, kap * ar * (a == b ? 2.0 : 1.0) / 12.0});
        }
      }
    }
  }
  for (int v : s->point_v)
    for (int i = 0; i < ns; ++i) bnd[(size_t)v * nf + i] += s->model.point_flux[i];
  // merge duplicates (stable order of first appearance => deterministic)
  std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.row != b.row ? a.row < b.row : a.col < b.col; });
  std::vector<int32_t> rrow, rcol, rptr((size_t)s->ndof + 1, 0); std::vector<double> rval; std::vector<int64_t> raddr;
  for (size_t k = 0; k < ents.size();) {
    size_t j = k; double v = 0.0;
    while (j < ents.size() && ents[j].row == ents[k].row && ents[j].col == ents[k].col) v += ents[j++].v;
    const int row = ents[k].row, col = ents[k].col;
    const int I = row / nf, i = row % nf, J = col / nf, jf = col % nf;
    const int32_t* b = t.cols.data() + t.rowptr[I]; const int32_t* e = t.cols.data() + t.rowptr[I + 1];
    const int kpos = t.sellk[(int)(std::lower_bound(b, e, J) - t.cols.data())];
    const int sl = t.node_slice[I], il = I - t.slice_node0[sl];
    rrow.push_back(row); rcol.push_back(col); rval.push_back(v);
    raddr.push_back(t.slice_off[sl] + (int64_t)(kpos * nf + jf) * kWave + il * nf + i);
    rptr[row + 1]++;
    k = j;
  }
  for (int r = 0; r < s->ndof; ++r) rptr[r + 1] += rptr[r];
  HIP_TRY(s->bndF.upload(bnd)); HIP_TRY(s->robF_ptr.upload(rptr)); HIP_TRY(s->rob_col.upload(rcol));
  HIP_TRY(s->rob_row.upload(rrow)); HIP_TRY(s->rob_val.upload(rval)); HIP_TRY(s->rob_addr.upload(raddr));
  s->c.bndF = s->bndF.p; s->c.robF_ptr = s->robF_ptr.p; s->c.rob_col = s->rob_col.p; s->c.rob_row = s->rob_row.p;
  s->c.rob_val = s->rob_val.p; s->c.rob_addr = s->rob_addr.p; s->c.n_robin = (int)rval.size();
  return GMPNP_OK;
}

int check_model(const gmpnp_model_t* m, int dim) {
  if (!m) return fail(GMPNP_ERR_INVALID, "model is NULL");
  if (m->dim != dim) return fail(GMPNP_ERR_INVALID, "model.dim != mesh.dim");
  if (m->n_species < 1 || m->n_species > GMPNP_MAX_SPECIES) return fail(GMPNP_ERR_INVALID, "model.n_species out of range");
  if (m->n_bilinear < 0 || m->n_bilinear > GMPNP_MAX_BILINEAR) return fail(GMPNP_ERR_INVALID, "model.n_bilinear out of range");
  for (int t = 0; t < m->n_bilinear; ++t)
    if (m->bil_j[t] < 0 || m->bil_j[t] >= m->n_species || m->bil_k[t] < 0 || m->bil_k[t] >= m->n_species)
      return fail(GMPNP_ERR_INVALID, "model.bil_j/bil_k out of range");
  if (!(m->inv_dt == m->inv_dt)) return fail(GMPNP_ERR_INVALID, "model.inv_dt is NaN");
  return GMPNP_OK;
}

// ---- device passes --------------------------------------------------------------------------
template <int DIM, int NF>
int launch_element(gmpnp_solver* s, bool want_j) {
  const int g = grid_for(s->t.nc, 64);
  if constexpr (DIM == 3) {
    // large meshes: record stores staged through LDS (k_element<.., STAGED>); the reference meshes keep the direct form
    if (want_j && s->staged_element) {
      hipLaunchKernelGGL((k_element<DIM, NF, true, true>), dim3(g), dim3(64), 0, s->stream, s->c);
      HIP_TRY(hipGetLastError());
      return GMPNP_OK;
    }
  }
  if (want_j) hipLaunchKernelGGL((k_element<DIM, NF, true>), dim3(g), dim3(64), 0, s->stream, s->c);
  else hipLaunchKernelGGL((k_element<DIM, NF, false>), dim3(g), dim3(64), 0, s->stream, s->c);
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

template <int DIM, int NF>
int launch_res_gather(gmpnp_solver* s) {
  hipLaunchKernelGGL((k_res_gather<DIM, NF>), dim3(s->n_resblocks), dim3(kVecBlock), 0, s->stream, s->c);
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

template <int DIM, int NF>
int launch_jac_gather(gmpnp_solver* s) {
  const int g = grid_for(s->c.n_work * kWave, kVecBlock);   // n_work = 8 equal runs, each a whole number of workgroups
  hipLaunchKernelGGL((k_jac_gather<DIM, NF>), dim3(g), dim3(kVecBlock), 0, s->stream, s->c);
  if (s->c.n_robin > 0) hipLaunchKernelGGL(k_robin_add, dim3(grid_for(s->c.n_robin, 256)), dim3(256), 0, s->stream, s->c);
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

int drain_spmv_events(gmpnp_solver* s);

// residual at the current u: returns ||b||_2 and the device status flags
template <int DIM, int NF>
int residual(gmpnp_solver* s, bool want_j, double* norm, int* flags) {
  int rc = launch_element<DIM, NF>(s, want_j); if (rc) return rc;
  rc = launch_res_gather<DIM, NF>(s); if (rc) return rc;
  // the partials and the status word land in pinned host memory by the kernel's own stores: no copy in the stream
  HIP_TRY(hipStreamSynchronize(s->stream));
  rc = drain_spmv_events(s); if (rc) return rc;
  double acc = 0.0;
  for (int i = 0; i < s->n_resblocks; ++i) acc += s->h_part[i];
  *norm = std::sqrt(acc);
  *flags = *s->h_status;
  return GMPNP_OK;
}

// `refresh` = false keeps the previous Dinv and coarse inverse (any nonsingular block scaling and any coarse operator
// give a valid right preconditioner) and only re-scales the new matrix: the cheap path of a lagged preconditioner.
template <int NF>
void launch_coarse_chain(gmpnp_solver* s, const Ctx& c, hipStream_t st) {
  hipLaunchKernelGGL((k_coarse_rows<NF>), dim3(s->t.nslices), dim3(64), 0, st, c);
  hipLaunchKernelGGL((k_coarse_sum<NF>), dim3(s->t.nagg * s->c.coarse_chunks), dim3(kVecBlock), 0, st, c);
  const int n = s->ncoarse;
  hipLaunchKernelGGL(k_coarse_reduce, dim3(grid_for(n * n, kVecBlock)), dim3(kVecBlock), 0, st, c);
  hipLaunchKernelGGL((k_coarse_invert<NF>), dim3(1), dim3(512), coarse_lds_bytes(n, NF), st, c);
}

// `allow_async` (Newton): unless `refresh_coarse` demands an inverse of THIS matrix now, the coarse chain of this matrix
// is started on the side stream and the solve runs with the inverse the previous chain left (any coarse operator gives a
// valid preconditioner; one iteration of staleness costs < 1 % BiCGStab iterations, the chain is 155 us of a single
// stream otherwise).  The chain reads vals_s and owns AP / AcPart / Ac: it is awaited (in stream order, the host does
// not block) before the next k_scale_columns and before any chain in the main stream.
template <int DIM, int NF>
int ml_setup(gmpnp_solver* s);
template <int DIM, int NF>
int setup_preconditioner(gmpnp_solver* s, int mode, bool refresh = true, bool refresh_coarse = true, bool allow_async = false) {
  s->c.use_coarse = (mode == GMPNP_LINEAR_BICGSTAB_TWOLEVEL) ? 1 : 0;
  if (!s->precond_valid || s->precond_mode != mode) refresh = refresh_coarse = true;
  if constexpr (DIM == 3 && NF == 9) {
    if (s->ml_coarse && refresh) { int rc = ml_setup<DIM, NF>(s); if (rc) return rc; }   // level Jacobians at the injected state
  }
  if (refresh) hipLaunchKernelGGL((k_block_inverse<NF>), dim3(grid_for(s->t.nv, 4)), dim3(64), 0, s->stream, s->c);
  const bool had_chain = s->chain_in_flight;
  if (had_chain) { HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_chain, 0)); s->chain_in_flight = false; }
  hipLaunchKernelGGL((k_scale_columns<NF>), dim3(grid_for(s->c.n_work * kWave, kVecBlock)), dim3(kVecBlock), 0, s->stream, s->c);
  if (s->c.use_coarse && refresh) {
    const bool async = allow_async && s->coarse_async && s->stream2 != nullptr;
    if (async && !refresh_coarse) {
      if (had_chain) { s->aci_cur ^= 1; s->c.Aci = s->aci_buf[s->aci_cur]; }   // adopt what the previous chain left
      HIP_TRY(hipEventRecord(s->ev_mat, s->stream));
      HIP_TRY(hipStreamWaitEvent(s->stream2, s->ev_mat, 0));
      Ctx c2 = s->c; c2.Aci = s->aci_buf[s->aci_cur ^ 1];
      launch_coarse_chain<NF>(s, c2, s->stream2);
      HIP_TRY(hipEventRecord(s->ev_chain, s->stream2));
      s->chain_in_flight = true;
    } else if (refresh_coarse) {
      launch_coarse_chain<NF>(s, s->c, s->stream);   // inverse of this matrix, now (a pending side-stream result is dropped)
      if (async) {   // ... and the double buffer keeps rolling: nothing in flight, the next set-up starts a new chain
      }
    }
  }
  HIP_TRY(hipGetLastError());
  s->precond_valid = true; s->precond_mode = mode;
  return GMPNP_OK;
}

// ---- geometric multilevel term (gmpnp_multilevel.h); every launch goes to the FINEST handle's stream ------------------------------
template <int NF>
int apply_minv(gmpnp_solver* s, int mode, const double* src, double* dst, double scale_dst, double scale_x, const NewtonUpdate* upd = nullptr);
template <int NF>
int ml_level_apply(gmpnp_solver* top, gmpnp_solver* L);
// L->ml_r holds the restricted residual r of level L; L->ml_w = S_L r, one V(1,1) cycle from x = 0 (a FIXED linear operator, as
// BiCGStab requires) with the level's own two-level preconditioner M_L^-1 = Dinv_L (I + Ps Aci Ps^T) as the smoother:
//     x  = w M_L^-1 r                                       pre-smoothing (w = ml_omega, damped)
//     x += mask P S_{L+1} mask P^T (r - J_L x)              coarse-grid correction, if a coarser level is attached
//     x += w M_L^-1 (r - J_L x)                             post-smoothing; the COARSEST level repeats it ml_sweeps - 1 times
// Two SpMVs with the level's Jacobian per cycle (1/8 of the next finer level's each).  Scratch: the level handle's Krylov vectors
// (it never runs a solve of its own).
template <int NF>
int ml_level_apply(gmpnp_solver* top, gmpnp_solver* L) {
  hipStream_t st = top->stream, keep = L->stream;
  const int n = (int)L->ndof;
  const dim3 vg(grid_for(n, 256));
  auto smooth = [&](const double* src, double scale_dst) -> int {   // ml_w = scale_dst * ml_w + omega * M_L^-1 src
    if (L->ml_coarse && L->ml_mid_jacobi) {   // intermediate level: damped node-block Jacobi alone, one launch
      hipLaunchKernelGGL((k_ml_jacobi<NF>), vg, dim3(256), 0, st, (const double*)L->c.Dinv, src, L->ml_w.p, scale_dst, L->ml_omega, n);
      return GMPNP_OK;
    }
    L->stream = st;
    const int rc = apply_minv<NF>(L, GMPNP_LINEAR_BICGSTAB_TWOLEVEL, src, L->ml_w.p, scale_dst, L->ml_omega, nullptr);
    L->stream = keep;
    return rc;
  };
  auto residual = [&]() {   // ks = r - J_L ml_w, one launch
    hipLaunchKernelGGL((k_spmv_residual<NF>), dim3(L->t.own_ntiles), dim3(kKrylovThreads), 0, st, L->c, (const double*)L->ml_w.p, (const double*)L->ml_r.p, L->ks.p);
  };
  int rc = smooth(L->ml_r.p, 0.0); if (rc) return rc;
  if (gmpnp_solver* C = L->ml_coarse) {
    residual();
    hipLaunchKernelGGL((k_ml_restrict<NF>), dim3(grid_for(C->ndof, 256)), dim3(256), 0, st, (const double*)L->ks.p, L->c.bcflag, (const int32_t*)L->ml_child_ptr.p,
                       (const int32_t*)L->ml_child.p, C->c.bcflag, C->ml_r.p, (int)C->ndof);
    rc = ml_level_apply<NF>(top, C); if (rc) return rc;
    hipLaunchKernelGGL((k_ml_prolong_add<NF>), vg, dim3(256), 0, st, (const double*)C->ml_w.p, (const int32_t*)L->ml_par.p, L->c.bcflag, L->ml_w.p, n);
#ifndef GMPNP_ML_NO_POST   // (A/B builds only: V(1,0) on the intermediate levels)
    residual();
    rc = smooth(L->ks.p, 1.0); if (rc) return rc;
#endif
  } else {
    for (int k = 1; k < L->ml_sweeps; ++k) { residual(); rc = smooth(L->ks.p, 1.0); if (rc) return rc; }
  }
  return GMPNP_OK;
}
// coarse part of T src on the finest level: leaves it in s->ml_coarse->ml_w (to be prolonged by the caller's kernel)
template <int NF>
int ml_correction(gmpnp_solver* s, const double* src) {
  gmpnp_solver* C = s->ml_coarse;
  hipLaunchKernelGGL((k_ml_restrict<NF>), dim3(grid_for(C->ndof, 256)), dim3(256), 0, s->stream, src, s->c.bcflag, (const int32_t*)s->ml_child_ptr.p,
                     (const int32_t*)s->ml_child.p, C->c.bcflag, C->ml_r.p, (int)C->ndof);
  return ml_level_apply<NF>(s, C);
}
// z = vec + theta D T vec: the operand a materialised half-iteration stages
template <int NF>
int ml_stage(gmpnp_solver* s, const double* vec) {
  int rc = ml_correction<NF>(s, vec); if (rc) return rc;
  hipLaunchKernelGGL((k_ml_stage<NF>), dim3(grid_for(s->ndof, kMlStageNodes * NF)), dim3(kMlStageNodes * NF), 0, s->stream, s->c, (const double*)s->ml_coarse->ml_w.p,
                     (const int32_t*)s->ml_par.p, vec, s->ml_z.p, s->ml_theta);
  return GMPNP_OK;
}
// Once per preconditioner set-up: the state goes one level down by injection, the coarser level assembles ITS Jacobian there
// (element kernel + gather of this library on the level's own mesh) and sets up its own two-level preconditioner — which, if a
// still coarser level is attached to it, does the same one level further down (setup_preconditioner calls this function).
template <int DIM, int NF>
int ml_setup(gmpnp_solver* s) {
  gmpnp_solver* C = s->ml_coarse;
  hipStream_t keep = C->stream;
  C->stream = s->stream;   // (the level handle's own stream stays unused: everything of the hierarchy is ordered in the finest handle's)
  hipLaunchKernelGGL((k_ml_inject<NF>), dim3(grid_for(C->ndof, 256)), dim3(256), 0, s->stream, (const double*)s->u.p, (const int32_t*)s->ml_copy.p, C->u.p, (int)C->ndof);
  int rc = launch_element<DIM, NF>(C, true);
  if (!rc) rc = launch_jac_gather<DIM, NF>(C);
  if (!rc) { C->jacobian_valid = true; rc = setup_preconditioner<DIM, NF>(C, GMPNP_LINEAR_BICGSTAB_TWOLEVEL, true, true, false); }
  C->stream = keep;
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

// One half-iteration of the fused BiCGStab: one launch (coarse workgroups inside the tile launch), two, or three.
template <int NF, int WHICH>
int launch_half(gmpnp_solver* s, int k) {
  const dim3 cg(std::max(1, s->t.nagg));
  if (s->fused_half) {
    const dim3 fg(s->t.nagg + s->t.own_ntiles);
    const unsigned target = (unsigned)(++s->fused_seq);
    if (WHICH == 0) hipLaunchKernelGGL((k_half_a<NF>), fg, dim3(kKrylovThreads), 0, s->stream, s->c, k, target);
    else hipLaunchKernelGGL((k_half_b<NF>), fg, dim3(kKrylovThreads), 0, s->stream, s->c, k, target);
  } else if (s->matp) {   // materialised vectors: coarse kernel, streaming vector update, tile kernel staging one vector
    const dim3 vg(grid_for(s->ndof, 256));
    // with a multilevel term the tile kernel stages vec + theta D T vec (ml_stage) instead of the vector k_vec_a / k_vec_b wrote
    if (WHICH == 0) {
      hipLaunchKernelGGL((k_coarse_a<NF>), cg, dim3(kCoarseThreads), 0, s->stream, s->c, k);
      hipLaunchKernelGGL(k_vec_a, vg, dim3(256), 0, s->stream, s->c, k);
      if (s->ml_coarse) {
        if constexpr (NF == 9) { int rc = ml_stage<NF>(s, s->c.kp[k & 1]); if (rc) return rc; }
        Ctx cc = s->c; cc.stage_a = s->ml_z.p;
        hipLaunchKernelGGL((k_bicg_a_mat<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, cc, k);
      } else hipLaunchKernelGGL((k_bicg_a_mat<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, k);
    } else {
      hipLaunchKernelGGL((k_coarse_b<NF>), cg, dim3(kCoarseThreads), 0, s->stream, s->c, k);
      hipLaunchKernelGGL(k_vec_b, vg, dim3(256), 0, s->stream, s->c, k);
      if (s->ml_coarse) {
        if constexpr (NF == 9) { int rc = ml_stage<NF>(s, s->c.ks); if (rc) return rc; }
        Ctx cc = s->c; cc.stage_b = s->ml_z.p;
        hipLaunchKernelGGL((k_bicg_b_mat<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, cc, k);
      } else hipLaunchKernelGGL((k_bicg_b_mat<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, k);
    }
  } else if (WHICH == 0) {
    hipLaunchKernelGGL((k_coarse_a<NF>), cg, dim3(kCoarseThreads), 0, s->stream, s->c, k);
    hipLaunchKernelGGL((k_bicg_a<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, k);
  } else {
    hipLaunchKernelGGL((k_coarse_b<NF>), cg, dim3(kCoarseThreads), 0, s->stream, s->c, k);
    hipLaunchKernelGGL((k_bicg_b<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, k);
  }
  if (s->prereduce) {   // the sums the next coarse kernel reads
    const int n = s->ncoarse;
    if (WHICH == 0) hipLaunchKernelGGL(k_dist_reduce, dim3(2 + 3 * n), dim3(256), 0, s->stream, s->c, 1, k & 1, s->red_a.p);
    else hipLaunchKernelGGL(k_dist_reduce, dim3(4 + n), dim3(256), 0, s->stream, s->c, 2, k & 1, s->red_b.p);
  }
  s->spmv_launched++;
  return GMPNP_OK;
}

int drain_spmv_events(gmpnp_solver* s) {
  for (size_t i = 0; i < s->ev_used; ++i) {
    if (s->ev_halves[i] <= 0) continue;   // a bracket that reached into the launches behind the end of its solve
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_pool[i].first, s->ev_pool[i].second));
    s->spmv_us_sum += 1000.0 * ms; s->spmv_sampled += s->ev_halves[i];
  }
  s->ev_used = 0;
  return GMPNP_OK;
}

// Iteration k of the solve (k = 0, 1, ...): the index is a kernel ARGUMENT, so no kernel has to read it back from
// memory before it can address its parity buffers.
template <int NF>
int enqueue_iteration(gmpnp_solver* s, int k) {
  int rc = launch_half<NF, 0>(s, k);
  if (rc) return rc;
  return launch_half<NF, 1>(s, k);
}

// Solve J dx = rhs (rhs already in c.kr on the device, ||rhs|| = bnorm) with the fused right-preconditioned BiCGStab;
// leaves y in c.ky; the caller applies M^{-1} (apply_minv).
template <int NF>
int krylov(gmpnp_solver* s, int mode, double bnorm, double rtol, double atol, int maxit, gmpnp_linear_stats_t* st, bool restart = false) {
  const int use_coarse = (mode == GMPNP_LINEAR_BICGSTAB_TWOLEVEL) ? 1 : 0;
  s->c.use_coarse = use_coarse;
  const int n = s->ndof;
  s->fused_seq = 0;
  KrylovScalars init{};
  init.rho[0] = init.rho[1] = s->shadow_src ? s->shadow_rho0 : bnorm * bnorm; init.alpha = 1.0;
  init.tol = std::max(rtol * bnorm, atol); init.rr = bnorm * bnorm; init.iters = 0; init.it_cur = 0;
  init.max_iters = maxit; init.done = 0; init.done_next = 0; init.omega = 0.0; init.beta = 0.0;
  init.rr0 = bnorm * bnorm;
  if (!(bnorm > 0.0)) init.done = 1;  // zero right-hand side: dx = 0
  // one launch: shadow vector (r_0, or a pseudo-random vector after a breakdown), y = 0, P^T r_0 partials
  // where A(0) expects them, hand-over flags cleared, scalars from the kernel argument
  hipLaunchKernelGGL((k_krylov_init<NF>), dim3(s->t.own_ntiles), dim3(kVecBlock), 0, s->stream, s->c, s->shadow_src, init, s->cpart_v1.p);
  if (s->prereduce && use_coarse) hipLaunchKernelGGL(k_dist_reduce, dim3(s->ncoarse), dim3(256), 0, s->stream, s->c, 0, 0, s->red_i.p);
  // the previous solve wrote its verdict before the host left its loop and nothing of it writes the mirror afterwards
  volatile HostPoll* hp = s->h_poll;
  hp->done = 0; hp->iters = 0; hp->rr = 0.0;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  const int B = s->burst_iters;
  // Bursts of B iterations.  The first burst is 3/4 of what the previous solve with this preconditioner
  // needed; after that the host polls the device flag one burst BEHIND the launches (copy + event, launch the
  // next burst, then wait for the event), so the read-back latency hides behind queued work.  Kernels of a
  // converged solve exit at their first instruction.
  // first burst: 7/8 of the count the same Newton iteration needed one time step ago (solves of one index resemble each
  // other far more than consecutive solves do: the first of a step is cold, the others are warm-started), else 3/4 of
  // the previous solve
  // With the pinned progress mirror the host keeps up one iteration at a time, so the first burst is insurance against
  // a slow host rather than a way to save polls: half the expected count (measured on the bench, sixteenths of the
  // hint: 0..8 -> 529-533 its/s, 12 -> 525, 14 -> 522, 16 -> 516; more surplus early-exit launches the longer it is).
  const int expect = s->krylov_hint > 0 ? s->krylov_hint / 2 : s->last_krylov_iters[use_coarse] / 2;
  int first = s->opts.krylov_batch > 0 ? s->opts.krylov_batch : std::max(B, expect);
  if (restart) first = B;  // a restart pass only has to remove the drift
  first = ((first + B - 1) / B) * B;
  int next_k = 0;  // iteration index of the next launch (the device stops advancing once `done` is set)
  auto burst = [&](int iters) -> int {
    for (int it = 0; it < iters; ++it) { int rc = enqueue_iteration<NF>(s, next_k++); if (rc) return rc; }
    return GMPNP_OK;
  };
  KrylovScalars res = init;
  int launched = 0, slot = 0;
  // Timing for the roofline (opts.profile_every = N > 0): ONE event pair around the first burst of every Nth solve — a run of
  // back-to-back half-iterations, all of them live (checked against the solve's iteration count afterwards).  Elapsed time /
  // half-iterations = what a half-iteration costs in the solve, launch gaps included.  (Events attached to single dispatches
  // read 1.5 us more than rocprofv3's kernel durations: a dispatch with a completion signal of its own is a slower dispatch.)
  const int every = s->opts.profile_every;
  long bracket = -1;
  if (every > 0 && !restart && !res.done && first >= 2 && (s->solve_index++ % every) == 0) {
    if (s->ev_used == s->ev_pool.size()) {
      hipEvent_t a, b; HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b)); s->ev_pool.push_back({a, b}); s->ev_halves.push_back(0);
    }
    bracket = (long)s->ev_used++;
    s->ev_halves[bracket] = 0;
    HIP_TRY(hipEventRecord(s->ev_pool[bracket].first, s->stream));
  }
  if (!res.done) {
    int rc = burst(first); if (rc) return rc;
    if (bracket >= 0) HIP_TRY(hipEventRecord(s->ev_pool[bracket].second, s->stream));
    launched = first;
    bool mirror_ok = s->host_poll != 0;
    while (mirror_ok) {
      // The B kernels report progress straight into pinned host memory: launch the next burst, then spin until the
      // iterations launched BEFORE it are done (or the solve is).  No copy kernel and no event in the stream.
      const int target = launched;
      rc = burst(B); if (rc) return rc;
      launched += B;
      const double t_spin = now_ms();
      int spins = 0;
      while (!hp->done && hp->iters < target) {
        __builtin_ia32_pause();
        if ((++spins & 0xfff) == 0 && now_ms() - t_spin > 5000.0) { mirror_ok = false; break; }  // GPU stuck or mirror not visible
      }
      if (!mirror_ok) break;
      if (hp->done) {
        std::atomic_thread_fence(std::memory_order_acquire);
        res.done = hp->done; res.iters = hp->iters; res.rr = hp->rr;
        break;
      }
      if (launched > maxit + 4 * B) { mirror_ok = false; break; }  // defensive: the device test ends the loop at max_iters
    }
    while (!mirror_ok) {
      HIP_TRY(hipMemcpyAsync(&s->h_scal[slot], s->scal.p, sizeof(KrylovScalars), hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipEventRecord(s->ev_poll[slot], s->stream));
      rc = burst(B); if (rc) return rc;  // speculative: overlaps the read-back
      launched += B;
      HIP_TRY(hipEventSynchronize(s->ev_poll[slot]));
      res = s->h_scal[slot];
      slot ^= 1;
      if (res.done) break;
      if (launched > maxit + 4 * B) break;  // defensive: the device test ends the loop at max_iters
    }
    // the sampled launches' events are read at the next point where the stream is synchronised anyway (residual(),
    // the end of gmpnp_linear_solve, gmpnp_spmv_profile): no wait of its own
  }
  if (bracket >= 0 && res.done == 1 && res.iters > first) s->ev_halves[bracket] = 2 * first;   // every launch of the burst did its work
  if (!restart) s->last_krylov_iters[use_coarse] = res.iters;
  s->last_done = res.done;
  if (st) { st->iterations = res.iters; st->converged = (res.done == 1); st->residual_norm = std::sqrt(res.rr); st->rhs_norm = bnorm; }
  if (res.done != 1) {
    char buf[160];
    snprintf(buf, sizeof buf, "BiCGStab stopped without convergence (code %d) after %d iterations, ||r|| = %.3e, ||b|| = %.3e",
             res.done, res.iters, std::sqrt(res.rr), bnorm);
    return fail(GMPNP_ERR_LINEAR, buf);
  }
  return GMPNP_OK;
}

// dst = scale_dst*dst + scale_x * M^{-1} src,  M^{-1} = Dinv (I + P Aci P^T)
template <int NF>
int apply_minv(gmpnp_solver* s, int mode, const double* src, double* dst, double scale_dst, double scale_x,
               const NewtonUpdate* upd) {
  s->c.use_coarse = (mode == GMPNP_LINEAR_BICGSTAB_TWOLEVEL) ? 1 : 0;
  const bool pre = s->prereduce && s->c.use_coarse;
  if (s->c.use_coarse)
    hipLaunchKernelGGL((k_restrict<NF>), dim3(s->t.own_ntiles), dim3(kVecBlock), 0, s->stream, s->c, src, s->cpart_v0.p);
  if (pre) hipLaunchKernelGGL(k_dist_reduce, dim3(s->ncoarse), dim3(256), 0, s->stream, s->c, 3, 0, s->red_i.p);
  hipLaunchKernelGGL((k_minv_apply<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, src,
                     (const double*)s->cpart_v0.p, dst, scale_dst, scale_x, upd ? *upd : NewtonUpdate{nullptr, nullptr, 0.0, 0.0, 0.0},
                     pre ? (const double*)s->red_i.p : (const double*)nullptr);
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

// Krylov solve with the answer checked: BiCGStab stops on its RECURSIVE residual, which can drift away from
// b - J dx over thousands of iterations (plain Jacobi mode on stiff systems).  After each pass the true residual is
// formed with the unscaled matrix; if it misses the target grossly (> 1000x), the solve restarts on it (dx accumulates
// in kx).  stats->residual_norm reports the TRUE residual.
// rhs in c.kr on entry; dx = kx on return.
// kr = kb - J kx with the unscaled matrix; returns ||kr|| (synchronises the stream)
template <int NF>
int true_residual(gmpnp_solver* s, double* rn) {
  hipLaunchKernelGGL((k_spmv_plain<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, (const double*)s->kx.p, s->kt.p);
  hipLaunchKernelGGL(k_true_residual, dim3(s->n_resblocks), dim3(kVecBlock), 0, s->stream, (const double*)s->kb.p, (const double*)s->kt.p,
                     s->kr.p, s->c.part_f, (int)s->ndof);
  HIP_TRY(hipStreamSynchronize(s->stream));
  double acc = 0.0;
  for (int i = 0; i < s->n_resblocks; ++i) acc += s->h_part[i];
  *rn = std::sqrt(acc);
  return GMPNP_OK;
}

// `warm_scale` != 0: kx holds the previous Newton correction; the solve starts from x0 = warm_scale * kx, i.e. BiCGStab
// only has to remove b - J x0.  With the reference's damped update (omega = 0.9) consecutive corrections satisfy
// dx_{k+1} = (1 - omega) dx_k + O(|dx_k|^2), so x0 = (1 - omega) dx_k leaves a second-order small residual and the
// same absolute target is reached in far fewer iterations.  Falls back to x0 = 0 when x0 does not reduce the residual.
template <int NF>
int krylov_verified(gmpnp_solver* s, int mode, double bnorm, double rtol, double atol, int maxit, gmpnp_linear_stats_t* st,
                    int verify_above = 0, double warm_scale = 0.0, double warm_prev = 0.0, const double* rhs_src = nullptr,
                    bool rhs_ready = false, bool x0_ready = false, const NewtonUpdate* upd = nullptr, bool* upd_done = nullptr,
                    bool dots_in_flight = false) {
  const int n = s->ndof;
  if (!std::isfinite(bnorm)) return fail(GMPNP_ERR_LINEAR, "right-hand side of the linear system is not finite");
  const double tol = std::max(rtol * bnorm, atol);
  // kb keeps the right-hand side; `rhs_src` (Newton: F) saves the caller's separate copy into kr
  // `rhs_ready`: k_res_gather already left b in kr and kb (Newton)
  if (rhs_ready) {}
  else if (rhs_src) hipLaunchKernelGGL(k_copy2, dim3(grid_for(n, 256)), dim3(256), 0, s->stream, s->kr.p, s->kb.p, rhs_src, n);
  else HIP_TRY(hipMemcpyAsync(s->kb.p, s->kr.p, n * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
  if (upd_done) *upd_done = false;
  gmpnp_linear_stats_t total{}; total.rhs_norm = bnorm;
  double rhs_norm = bnorm;
  bool warm = false;
  if (warm_scale != 0.0 && bnorm > 0.0) {
    // x0 = warm_scale * kx + warm_prev * kxp (left in kx by the previous Newton update when `x0_ready`); accepted when
    // it removes at least half of the residual: one plain SpMV, three dots, one host round trip
    if (!x0_ready) hipLaunchKernelGGL(k_warm_start, dim3(grid_for(n, 256)), dim3(256), 0, s->stream, s->kx.p, s->kxp.p, warm_scale, warm_prev, n);
    if (dots_in_flight) {   // w = J x0 and the dot products were started on the side stream right after the Jacobian gather
      HIP_TRY(hipEventSynchronize(s->ev_dots));
      HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_dots, 0));   // kt is read by k_start_residual below
    } else {
      hipLaunchKernelGGL((k_spmv_plain<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, (const double*)s->kx.p, s->kt.p);
      hipLaunchKernelGGL(k_dots3, dim3(s->n_resblocks), dim3(kVecBlock), 0, s->stream, (const double*)s->kt.p, (const double*)s->kb.p,
                         s->c.part_f, n, s->n_resblocks);
      HIP_TRY(hipStreamSynchronize(s->stream));
    }
    double wb = 0.0, ww = 0.0, bb = 0.0;
    for (int i = 0; i < s->n_resblocks; ++i) { wb += s->h_part[i]; ww += s->h_part[s->n_resblocks + i]; bb += s->h_part[2 * s->n_resblocks + i]; }
    // The predicted correction is taken as it is.  (Scaling it by the minimal-residual multiple (w,b)/(w,w) makes |r0|
    // smaller every time and BiCGStab slower all the same: 18.8k instead of 17.3k iterations over the bench, round 1.)
    const double rn2 = bb - 2.0 * wb + ww;  // ||b - w||^2
    if (rn2 == rn2 && rn2 >= 0.0 && rn2 < 0.25 * bb) {
      hipLaunchKernelGGL(k_start_residual, dim3(grid_for(n, 256)), dim3(256), 0, s->stream, s->kr.p, (const double*)s->kb.p,
                         (const double*)s->kt.p, n);
      warm = true; rhs_norm = std::sqrt(rn2);
    }  // else: kr still holds b, cold start
  }
  // Restarted BiCGStab.  A pass runs at most `restart_every` iterations; then (and after a breakdown) the true residual
  // b - J x is formed with the unscaled matrix and the next pass starts from it with a fresh shadow vector.  This bounds
  // the drift of the recursive residual, and it is what keeps BiCGStab from blowing up on the stiff systems of the
  // thinnest pores (L_50_R_1: ||r|| reached 1e53 inside 10,000 unrestarted iterations, the reference's direct solver
  // sails through).  A pass that ends in a breakdown or in a residual 1e5 times its start (device test) is thrown away
  // and repeated with a pseudo-random shadow vector.  Short solves (the normal case) take exactly one pass and no check.
  constexpr int restart_every = 1000;
  bool have_x = warm;      // kx holds a partial solution
  bool random_shadow = false;
  int bad_passes = 0;
  for (int pass = 0;; ++pass) {
    gmpnp_linear_stats_t ls{};
    int rc = GMPNP_OK;
    const int cap = std::min(restart_every, maxit - total.iterations);
    if (rhs_norm <= tol) { ls.converged = 1; ls.residual_norm = rhs_norm; s->last_done = 1; }  // nothing left to do
    else {
      if (random_shadow) {  // (rhat, r0) of the new shadow vector
        hipLaunchKernelGGL(k_fill_hash, dim3(grid_for(n, 256)), dim3(256), 0, s->stream, s->krand.p, (unsigned)(pass * 2654435761u), n);
        hipLaunchKernelGGL(k_dots3, dim3(s->n_resblocks), dim3(kVecBlock), 0, s->stream, (const double*)s->krand.p, (const double*)s->kr.p,
                           s->c.part_f, n, s->n_resblocks);
        HIP_TRY(hipStreamSynchronize(s->stream));
        double wb = 0.0;
        for (int i = 0; i < s->n_resblocks; ++i) wb += s->h_part[i];
        s->shadow_src = s->krand.p; s->shadow_rho0 = wb;
      }
      const bool first_cold = (pass == 0 && !warm);
      rc = krylov<NF>(s, mode, rhs_norm, first_cold ? rtol : 0.0, first_cold ? atol : tol, cap, &ls, pass > 0);
      s->shadow_src = nullptr;
    }
    total.iterations += ls.iterations; total.converged = ls.converged; total.residual_norm = ls.residual_norm;
    if (rc && rc != GMPNP_ERR_LINEAR) { if (st) *st = total; return rc; }
    const bool usable = (s->last_done == 1 || s->last_done == 2);  // converged or cap reached: y is a valid partial solution
    if (usable && ls.iterations > 0) {
      // the normal end of a solve inside Newton (first pass, converged, short enough to go unchecked): the final
      // M^-1 application also applies the Newton update and leaves the predicted start of the next solve
      const bool final_now = upd && pass == 0 && s->last_done == 1 && ls.iterations <= verify_above && bnorm > 0.0 && !s->ml_coarse;
      int rc2 = apply_minv<NF>(s, mode, s->ky.p, s->kx.p, have_x ? 1.0 : 0.0, 1.0, final_now ? upd : nullptr); if (rc2) return rc2;
      if constexpr (NF == 9) {
        if (s->ml_coarse) {   // x += theta T y (the multilevel term of M^-1 applied to the Krylov solution)
          rc2 = ml_correction<NF>(s, s->ky.p); if (rc2) return rc2;
          hipLaunchKernelGGL((k_ml_add_solution<NF>), dim3(grid_for(s->ndof, 256)), dim3(256), 0, s->stream, s->c, (const double*)s->ml_coarse->ml_w.p,
                             (const int32_t*)s->ml_par.p, s->kx.p, s->ml_theta);
        }
      }
      if (final_now && upd_done) *upd_done = true;
      have_x = true; random_shadow = false;
    } else if (!usable) {
      random_shadow = true; ++bad_passes;
      HIP_TRY(hipMemsetAsync(s->status.p, 0, sizeof(int32_t), s->stream));
      if (!have_x) HIP_TRY(hipMemcpyAsync(s->kr.p, s->kb.p, n * sizeof(double), hipMemcpyDeviceToDevice, s->stream));  // kr was the work vector
    }
    if (!have_x && s->last_done == 1) {  // zero right-hand side
      HIP_TRY(hipMemsetAsync(s->kx.p, 0, n * sizeof(double), s->stream)); have_x = true;
    }
    if (!(bnorm > 0.0)) break;
    if (pass == 0 && s->last_done == 1 && ls.iterations <= verify_above) break;  // short solves do not drift
    double rn = rhs_norm;
    if (have_x) { int rc2 = true_residual<NF>(s, &rn); if (rc2) return rc2; }  // kr = b - J x
    total.residual_norm = rn;
    // Converged by the recurrence and within 1000x of the target by the true residual: accepted.  (A 1e-10 solve of a
    // small right-hand side ends at the attainable accuracy of b - J dx in fp64, 3-30x the target late in a run.)
    // A non-finite true residual is a failed solve whatever the recurrence reported: the caller's direct fallback takes
    // over (3D) and nothing of this x reaches u.
    const bool finite = std::isfinite(rn);
    if (s->last_done == 1 && finite && rn <= 1e3 * tol) break;
    if (total.iterations >= maxit || bad_passes > 8 || !finite) {
      if (st) { total.converged = 0; *st = total; }
      char buf[200];
      snprintf(buf, sizeof buf, "BiCGStab stopped without convergence after %d iterations in %d passes (%d breakdowns), ||b - J x|| = %.3e, ||b|| = %.3e",
               total.iterations, pass + 1, bad_passes, rn, bnorm);
      return fail(GMPNP_ERR_LINEAR, buf);
    }
    rhs_norm = rn;  // next pass solves J ddx = r (already in kr)
  }
  if (st) *st = total;
  return GMPNP_OK;
}

// ---- 1D direct solver ---------------------------------------------------------------------------
int build_tridiagonal(gmpnp_solver* s) {
  const Topology& t = s->t; const int nv = t.nv, nf = s->nf;
  s->tri_ok = false;
  if (s->dim != 1) return GMPNP_OK;
  std::vector<int32_t> kp((size_t)nv * 3, -1);
  for (int I = 0; I < nv; ++I)
    for (int k = t.rowptr[I]; k < t.rowptr[I + 1]; ++k) {
      const int J = t.cols[k];
      if (J == I - 1) kp[(size_t)I * 3] = t.sellk[k];
      else if (J == I) kp[(size_t)I * 3 + 1] = t.sellk[k];
      else if (J == I + 1) kp[(size_t)I * 3 + 2] = t.sellk[k];
      else return GMPNP_OK;  // internal order is not the path order: direct solver unavailable
    }
  std::vector<int> ns; for (int n = nv;; n = (n + 1) / 2) { ns.push_back(n); if (n == 1) break; }
  size_t total = 0;
  for (int n : ns) total += (size_t)n * (5 * nf * nf + 3 * nf);
  HIP_TRY(s->tri_store.alloc(total));
  HIP_TRY(s->tri_kpos.upload(kp));
  double* p = s->tri_store.p;
  s->tri.clear();
  for (int n : ns) {
    TriLevel l{}; l.n = n;
    auto take = [&](size_t cnt) { double* q = p; p += cnt; return q; };
    l.L = take((size_t)nf * nf * n); l.D = take((size_t)nf * nf * n); l.U = take((size_t)nf * nf * n); l.b = take((size_t)nf * n);
    l.Li = take((size_t)nf * nf * n); l.Ui = take((size_t)nf * nf * n); l.bi = take((size_t)nf * n); l.x = take((size_t)nf * n);
    s->tri.push_back(l);
  }
  s->tri_ok = true;
  return GMPNP_OK;
}

// Solve J x = rhs (device pointer, internal order) by block cyclic reduction; x stays in tri[0].x (SoA).
template <int NF>
int tri_solve(gmpnp_solver* s, const double* rhs) {
  if (!s->tri_ok) return fail(GMPNP_ERR_INVALID, "block-tridiagonal solver needs a 1D mesh in path order");
  hipLaunchKernelGGL((k_tri_extract<NF>), dim3(grid_for(s->t.nv * NF * NF, kVecBlock)), dim3(kVecBlock), 0, s->stream,
                     s->c, s->tri[0], s->tri_kpos.p, rhs);
  const int nl = (int)s->tri.size();
  // the top of the pyramid (every level whose upper neighbour has at most kBcrTailRows rows) is ONE launch: k_bcr_tail
  int l0 = nl - 1;
  while (l0 > 0 && s->tri[l0].n <= kBcrTailRows && nl - l0 < kBcrTailLevels) --l0;   // tail = levels l0 .. nl-1 (tri[l0+1].n <= kBcrTailRows)
  for (int l = 0; l < l0; ++l)
    hipLaunchKernelGGL((k_bcr_forward<NF>), dim3(grid_for(s->tri[l + 1].n, 4)), dim3(64), 0, s->stream, s->tri[l],
                       s->tri[l + 1], s->status.p);
  {
    TriTail tt{};
    tt.nlev = nl - l0;
    for (int l = l0; l < nl; ++l) tt.lv[l - l0] = s->tri[l];
    hipLaunchKernelGGL((k_bcr_tail<NF>), dim3(1), dim3(kBcrTailThreads), 0, s->stream, tt, s->status.p);
  }
  for (int l = l0 - 1; l >= 0; --l)
    hipLaunchKernelGGL((k_bcr_backward<NF>), dim3(grid_for(s->tri[l].n * NF, kVecBlock)), dim3(kVecBlock), 0, s->stream,
                       s->tri[l], s->tri[l + 1]);
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

template <int NF>
int tri_apply(gmpnp_solver* s, double* dst, double scale_dst, double scale_x) {
  hipLaunchKernelGGL((k_tri_apply<NF>), dim3(grid_for(s->ndof, kVecBlock)), dim3(kVecBlock), 0, s->stream, s->tri[0], dst,
                     scale_dst, scale_x, s->ndof);
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

// ---- 3D direct solver: block-banded LU ------------------------------------------------------------
template <int NF>
int band_prepare(gmpnp_solver* s) {
  if (s->lu_ready) return GMPNP_OK;
  const Topology& t = s->t;
  const int n = t.nv, b = std::max(t.lu_band, kBandPanel - 1);   // the substitution's panel triangle lies inside the stored band
  const double gb = (double)n * (2.0 * b + 1.0) * NF * NF * sizeof(double) / 1e9;
  char buf[200];
  if (gb > s->lu_max_gb) {
    snprintf(buf, sizeof buf, "block-banded LU needs %.1f GB (%d node blocks, band %d), above gmpnp_options_t.band_lu_max_gb = %.1f", gb, n, b, s->lu_max_gb);
    return fail(GMPNP_ERR_INVALID, buf);
  }
  HIP_TRY(s->lu_band.alloc((size_t)n * (2 * b + 1) * NF * NF, false));
  HIP_TRY(s->lu_dinv.alloc((size_t)n * NF * NF)); HIP_TRY(s->lu_y.alloc((size_t)n * NF));
  HIP_TRY(s->lu_pos.upload(t.lu_pos)); HIP_TRY(s->lu_node.upload(t.lu_node));
  HIP_TRY(s->lu_part.alloc((size_t)kBandPanel * kBandChunks * NF));
  s->lu = BandLU{s->lu_band.p, s->lu_dinv.p, s->lu_pos.p, s->lu_node.p, n, b};
  s->lu_ready = true;
  return GMPNP_OK;
}

// factorise the assembled Jacobian (c.vals)
template <int NF>
int band_factor(gmpnp_solver* s) {
  int rc = band_prepare<NF>(s); if (rc) return rc;
  const BandLU& lu = s->lu;
  constexpr int G = kBandThreads / (NF * NF);
  HIP_TRY(hipMemsetAsync(lu.band, 0, s->lu_band.n * sizeof(double), s->stream));
  hipLaunchKernelGGL((k_band_scatter<NF>), dim3(s->t.nslices), dim3(64), 0, s->stream, s->c, lu);
  hipLaunchKernelGGL((k_band_step<NF>), dim3(1, 1), dim3(kBandThreads), 0, s->stream, lu, -1, s->status.p);
  for (int k = 0; k + 1 < lu.n; ++k) {
    const int w = std::min(lu.b, lu.n - 1 - k);
    hipLaunchKernelGGL((k_band_step<NF>), dim3(grid_for(w, kBandColChunk), grid_for(w, G) + 1), dim3(kBandThreads), 0, s->stream, lu, k, s->status.p);
  }
  HIP_TRY(hipGetLastError());
  s->direct_solves++;
  return GMPNP_OK;
}

template <int NF>
int band_substitute(gmpnp_solver* s, const double* rhs, double* x) {
  const BandLU& lu = s->lu;
  const int n = lu.n, P = kBandPanel;
  double* y = s->lu_y.p; double* part = s->lu_part.p;
  hipLaunchKernelGGL((k_band_gather<NF>), dim3((n * NF + 255) / 256), dim3(256), 0, s->stream, lu, rhs, y);
  for (int K0 = 0; K0 < n; K0 += P) {   // forward: panels in elimination order
    const int K1 = std::min(K0 + P, n);
    hipLaunchKernelGGL((k_band_panel<NF, true>), dim3(K1 - K0, kBandChunks), dim3(NF * kWave), 0, s->stream, lu, (const double*)y, part, K0, K1);
    hipLaunchKernelGGL((k_band_tri<NF, true>), dim3(1), dim3(NF * kWave), 0, s->stream, lu, y, (const double*)part, x, K0, K1);
  }
  for (int K0 = ((n - 1) / P) * P; K0 >= 0; K0 -= P) {   // backward: the same panels in reverse
    const int K1 = std::min(K0 + P, n);
    hipLaunchKernelGGL((k_band_panel<NF, false>), dim3(K1 - K0, kBandChunks), dim3(NF * kWave), 0, s->stream, lu, (const double*)y, part, K0, K1);
    hipLaunchKernelGGL((k_band_tri<NF, false>), dim3(1), dim3(NF * kWave), 0, s->stream, lu, y, (const double*)part, x, K0, K1);
  }
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

// Direct solve of J dx = b with b in kb: dx in kx, checked and refined with the true residual (the pivoting is
// restricted to the node blocks).  stats->residual_norm = ||b - J dx||.
// `as_direct` (the Newton loop): an answer that is finite but misses the tolerance is RETURNED, not refused — the reference's
// MUMPS / UMFPACK hand back whatever the factorisation gives (3D:792), and it is Newton's own residual test that judges the step;
// st->converged = 0 says so.  The parity hook gmpnp_linear_solve keeps the strict verdict.
template <int NF>
int band_solve(gmpnp_solver* s, double bnorm, double rtol, double atol, gmpnp_linear_stats_t* st, bool as_direct = false) {
  const double tol = std::max(rtol * bnorm, atol);
  int rc = band_factor<NF>(s); if (rc) return rc;
  rc = band_substitute<NF>(s, s->kb.p, s->kx.p); if (rc) return rc;
  double rn = 0.0, best = 0.0;
  rc = true_residual<NF>(s, &rn); if (rc) return rc;   // kr = b - J kx
  HIP_TRY(hipMemcpy(s->h_status, s->status.p, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (*s->h_status & 2) return fail(GMPNP_ERR_LINEAR, "block-banded LU: singular pivot block");
  for (int round = 0; round < 3 && rn == rn && rn > tol; ++round) {   // iterative refinement
    best = rn;
    rc = band_substitute<NF>(s, s->kr.p, s->ks.p); if (rc) return rc;
    hipLaunchKernelGGL(k_axpy, dim3(grid_for(s->ndof, 256)), dim3(256), 0, s->stream, s->kx.p, (const double*)s->ks.p, 1.0, (int)s->ndof);
    rc = true_residual<NF>(s, &rn); if (rc) return rc;
    if (!(rn < 0.5 * best)) break;   // attainable accuracy reached
  }
  if (st) { st->converged = (rn == rn && rn <= 1e3 * tol) ? 1 : 0; st->residual_norm = rn; st->rhs_norm = bnorm; }
  if (as_direct && rn == rn && !std::isinf(rn)) return GMPNP_OK;
  if (!(rn == rn) || rn > 1e3 * tol) {
    char buf[200];
    snprintf(buf, sizeof buf, "block-banded LU: ||b - J x|| = %.3e after refinement, ||b|| = %.3e", rn, bnorm);
    return fail(GMPNP_ERR_LINEAR, buf);
  }
  return GMPNP_OK;
}

std::string status_message(int flags) {
  std::string m;
  if (flags & 1) m += "1 - sum_j a_j u_j <= 0 at a quadrature point; ";
  if (flags & 2) m += "singular diagonal node block; ";
  if (flags & 4) m += "singular coarse operator; ";
  if (flags & 8) m += "in-launch hand-over timed out; ";
  return m;
}

template <int DIM, int NF>
int newton(gmpnp_solver* s, const gmpnp_newton_options_t& o, gmpnp_newton_stats_t& st) {
  const double t0 = now_ms();
  HIP_TRY(hipMemsetAsync(s->status.p, 0, sizeof(int32_t), s->stream));
  double r = 0.0; int flags = 0;
  double ta = now_ms();
  // Every residual evaluation also leaves the element Jacobian records (k_element<.., true>, 43 us instead of 27): the
  // state only changes in the update, so the records of the convergence test ARE those of the next iteration's
  // Jacobian, and the separate element pass per iteration (another 43 us) is gone.  Wasted only on the last test of a solve.
  int rc = residual<DIM, NF>(s, true, &r, &flags); if (rc) return rc;
  st.ms_assemble += now_ms() - ta;
  if (flags & 1) { st.steric_excursion = 1; if (s->strict_steric) return fail(GMPNP_ERR_NUMERIC, status_message(flags)); }
  const double r0 = r;
  st.residuals[0] = r; st.n_residuals = 1;
  auto conv = [&](double res) {
    if (!(res == res)) return false;
    const double rel = res / r0;  // 0/0 = NaN compares false, as in DOLFIN
    return rel < o.relative_tolerance || res < o.absolute_tolerance;
  };
  bool done = conv(r);
  if (!(r == r)) return fail(GMPNP_ERR_NUMERIC, "residual is NaN before the first Newton iteration");
  while (!done && st.iterations < o.maximum_iterations) {
    if (s->phase_timing) HIP_TRY(hipEventRecord(s->ev_phase[0], s->stream));
    rc = launch_jac_gather<DIM, NF>(s); if (rc) return rc;   // element records: left by the last residual evaluation
    s->jacobian_valid = true;
    if (s->phase_timing) HIP_TRY(hipEventRecord(s->ev_phase[1], s->stream));
    if (o.linear_solver == GMPNP_LINEAR_BLOCK_TRIDIAGONAL) {
      if (s->phase_timing) HIP_TRY(hipEventRecord(s->ev_phase[2], s->stream));
      if constexpr (DIM == 1) {
        rc = tri_solve<NF>(s, s->F.p); if (rc) return rc;
        rc = tri_apply<NF>(s, s->u.p, 1.0, -o.relaxation_parameter); if (rc) return rc;
      } else {
        return fail(GMPNP_ERR_INVALID, "block-tridiagonal solver needs a 1D mesh");
      }
    } else if (o.linear_solver == GMPNP_LINEAR_BAND_LU || s->direct_sticky > 0) {
      if constexpr (DIM == 3) {
        HIP_TRY(hipMemcpyAsync(s->kb.p, s->F.p, s->ndof * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        if (s->phase_timing) HIP_TRY(hipEventRecord(s->ev_phase[2], s->stream));
        gmpnp_linear_stats_t ls{};
        rc = band_solve<NF>(s, r, o.krylov_relative_tolerance, o.krylov_absolute_tolerance, &ls, true); if (rc) return rc;
        st.direct_solves++; s->x0_predicted = false;
        hipLaunchKernelGGL(k_axpy, dim3(grid_for(s->ndof, 256)), dim3(256), 0, s->stream, s->u.p, (const double*)s->kx.p,
                           -o.relaxation_parameter, (int)s->ndof);
      } else {
        return fail(GMPNP_ERR_INVALID, "block-banded LU is the 3D direct solver (1D meshes: GMPNP_LINEAR_BLOCK_TRIDIAGONAL)");
      }
    } else {
      // The coarse inverse is reused for up to coarse_lag Newton iterations, unless the state was just set from outside
      // (first solve of a run: the Jacobian changes a lot between iterations) or the last reuse cost iterations.
      // Asynchronous scheme (default): every iteration starts the coarse chain of its matrix on the side stream and solves
      // with the inverse of the previous one; an inverse of THIS matrix is only built in-stream when it has to be.
      // x0 = (1-w) dx_k + (1-w)^2 (dx_k - (1-w) dx_{k-1}): first-order prediction plus the second-order term observed
      // one iteration earlier, scaled by (1-w)^2 as the quadratic form scales (gmpnp_options_t.warm_start = 1: first order only)
      const double q = 1.0 - o.relaxation_parameter;
      double wa = 0.0, wb = 0.0;
      if (s->warm_start && st.iterations > 0 && q != 0.0) { wa = q; if (s->warm_start > 1 && st.iterations > 1) { wa = q + q * q; wb = -q * q * q; } }
      // (The previous time step's total update is useless as a start of a step's FIRST solve: optimal multiple ~1e-5,
      // measured in round 1.)
      const bool x0_ready = s->x0_predicted && st.iterations > 0 && wa != 0.0;
      // The test of the predicted start (w = J x0 and three dot products, then a host decision) needs the new Jacobian
      // only: it runs on the side stream while the main stream builds the preconditioner, and the host waits for ITS
      // event, so neither the two kernels nor the round trip sit on the critical path.
      bool dots_in_flight = false;
      if (x0_ready && s->stream2 && s->warm_async) {
        HIP_TRY(hipEventRecord(s->ev_jac, s->stream));
        HIP_TRY(hipStreamWaitEvent(s->stream2, s->ev_jac, 0));
        hipLaunchKernelGGL((k_spmv_plain<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream2, s->c, (const double*)s->kx.p, s->kt.p);
        hipLaunchKernelGGL(k_dots3, dim3(s->n_resblocks), dim3(kVecBlock), 0, s->stream2, (const double*)s->kt.p, (const double*)s->kb.p,
                           s->c.part_f, (int)s->ndof, s->n_resblocks);
        HIP_TRY(hipEventRecord(s->ev_dots, s->stream2));
        dots_in_flight = true;
      }
      const bool must = s->state_jumped || s->coarse_refresh_due;
      const bool async_ok = s->coarse_async != 0 && DIM == 3;
      const bool coarse_fresh = async_ok ? must : (s->coarse_lag <= 1 || (st.iterations % s->coarse_lag) == 0 || must);
      rc = setup_preconditioner<DIM, NF>(s, o.linear_solver, true, coarse_fresh, async_ok); if (rc) return rc;
      // rhs = b (current residual vector F): copied into kr and kb by krylov_verified
      if (s->phase_timing) HIP_TRY(hipEventRecord(s->ev_phase[2], s->stream));
      gmpnp_linear_stats_t ls{};
      // inside Newton only long solves are checked: a short one does not drift, and Newton's own residual test sees
      // whatever is left
      s->krylov_hint = st.iterations < 32 ? s->hint_by_newton_it[st.iterations] : 0;
      // coefficients of the NEXT iteration's predicted start (same rule as wa, wb above, one iteration on); the
      // experiments that decide on host-side dot products of their own keep the separate kernels
      double na = 0.0, nb = 0.0;
      if (s->warm_start && q != 0.0) { na = q; if (s->warm_start > 1 && st.iterations + 1 > 1) { na = q + q * q; nb = -q * q * q; } }
      const NewtonUpdate upd{s->u.p, s->kxp.p, o.relaxation_parameter, na, nb};
      bool upd_done = false;
      rc = krylov_verified<NF>(s, o.linear_solver, r, o.krylov_relative_tolerance, o.krylov_absolute_tolerance,
                               o.krylov_maximum_iterations, &ls, 500, wa, wb, nullptr, /*rhs_ready=*/true,
                               x0_ready, na != 0.0 ? &upd : nullptr, &upd_done, dots_in_flight);
      if (dots_in_flight) HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_dots, 0));  // whatever path the solve took: kt and vals are free again
      // feedback: a reused coarse inverse that doubles the iteration count of the last fresh solve is dropped
      if (st.iterations < 32) s->hint_by_newton_it[st.iterations] = ls.iterations;
      s->krylov_hint = 0;
      if (rc == GMPNP_OK && st.iterations == 0) s->direct_backoff /= 2;  // BiCGStab works again
      if (coarse_fresh) { s->krylov_fresh_iters = ls.iterations; s->coarse_refresh_due = false; }
      else if (ls.iterations > 2 * s->krylov_fresh_iters + 10) s->coarse_refresh_due = true;
      if (st.iterations < GMPNP_MAX_NEWTON_HISTORY) st.krylov_per_iteration[st.iterations] = ls.iterations;
      st.krylov_iterations += ls.iterations;
      if constexpr (DIM == 3) {
        // The reference's linear solver is direct (MUMPS, 3D:792): a Krylov solve that does not converge is not an
        // error there.  The block-banded LU takes over for this system, the rest of this Newton solve and the
        // next few solves (kb still holds b).
        if (rc == GMPNP_ERR_LINEAR && s->direct_fallback) {
          const std::string why = g_err;
          HIP_TRY(hipMemsetAsync(s->status.p, 0, sizeof(int32_t), s->stream));
          gmpnp_linear_stats_t ds{};
          rc = band_solve<NF>(s, r, o.krylov_relative_tolerance, o.krylov_absolute_tolerance, &ds, true);
          if (rc) g_err = why + "; direct fallback: " + g_err;
          else {  // back off: 8, 16, ... 256 Newton solves before BiCGStab is tried again (a failed try costs ~0.25 s)
            st.direct_solves++; s->coarse_refresh_due = true;
            s->direct_backoff = std::min(256, std::max(8, 2 * s->direct_backoff));
            s->direct_sticky = s->direct_backoff;
          }
        }
      }
      if (rc) {
        HIP_TRY(hipMemcpy(s->h_status, s->status.p, sizeof(int32_t), hipMemcpyDeviceToHost));
        if (*s->h_status) g_err += " [" + status_message(*s->h_status) + "]";
        return rc;
      }
      // x <- x - omega dx (done by the solve's last kernel in the normal case)
      if (upd_done) s->x0_predicted = true;
      else if (na != 0.0) {
        hipLaunchKernelGGL(k_update_predict, dim3(grid_for(s->ndof, 256)), dim3(256), 0, s->stream, s->u.p, s->kx.p, s->kxp.p,
                           o.relaxation_parameter, na, nb, (int)s->ndof);
        s->x0_predicted = true;
      } else {
        hipLaunchKernelGGL(k_axpy, dim3(grid_for(s->ndof, 256)), dim3(256), 0, s->stream, s->u.p, (const double*)s->kx.p,
                           -o.relaxation_parameter, (int)s->ndof);
        s->x0_predicted = false;
      }
    }
    if (s->phase_timing) HIP_TRY(hipEventRecord(s->ev_phase[3], s->stream));
    st.iterations++;
    ta = now_ms();
    rc = residual<DIM, NF>(s, true, &r, &flags); if (rc) return rc;  // synchronises the stream
    if (s->phase_timing) {  // gmpnp_options_t.phase_timing: device time per phase (five event records and one more wait per iteration)
      float ms01 = 0.f, ms12 = 0.f, ms23 = 0.f, ms3 = 0.f;
      (void)hipEventElapsedTime(&ms01, s->ev_phase[0], s->ev_phase[1]);
      (void)hipEventElapsedTime(&ms12, s->ev_phase[1], s->ev_phase[2]);
      (void)hipEventElapsedTime(&ms23, s->ev_phase[2], s->ev_phase[3]);
      HIP_TRY(hipEventRecord(s->ev_phase[4], s->stream));
      HIP_TRY(hipEventSynchronize(s->ev_phase[4]));
      (void)hipEventElapsedTime(&ms3, s->ev_phase[3], s->ev_phase[4]);
      st.ms_assemble += ms01 + ms3;   // Jacobian assembly + next residual (device time)
      st.ms_setup += ms12;
      st.ms_krylov += ms23;
    }
    if (flags & 1) { st.steric_excursion = 1; if (s->strict_steric) return fail(GMPNP_ERR_NUMERIC, status_message(flags)); }
    if (flags & 14) return fail(GMPNP_ERR_LINEAR, status_message(flags));
    if (st.n_residuals < GMPNP_MAX_NEWTON_HISTORY) st.residuals[st.n_residuals++] = r;
    // NaN / Inf stay fatal (DOLFIN would iterate to its limit on a NaN residual and raise there)
    if (!(r == r) || std::isinf(r)) return fail(GMPNP_ERR_NUMERIC, (flags & 1) ? "residual became NaN / Inf after an iterate left the admissible set (1 - sum_j a_j u_j <= 0)" : "residual became NaN");
    done = conv(r);
  }
  s->state_jumped = false;
  if (s->direct_sticky > 0 && o.linear_solver != GMPNP_LINEAR_BAND_LU) s->direct_sticky--;
  st.converged = done ? 1 : 0;
  st.ms_total = now_ms() - t0;
  if (!done) return fail(GMPNP_ERR_NOT_CONVERGED, "Newton solver did not converge because maximum number of iterations reached");
  return GMPNP_OK;
}

int upload_vec(gmpnp_solver* s, const double* file_order, double* dev) {
  const int nf = s->nf, nv = s->t.nv;
  HIP_TRY(hipStreamSynchronize(s->stream));   // the staging buffer is free again
  for (int i = 0; i < nv; ++i) std::memcpy(&s->h_stage[(size_t)i * nf], &file_order[(size_t)s->t.perm[i] * nf], nf * sizeof(double));
  HIP_TRY(hipMemcpyAsync(dev, s->h_stage, (size_t)s->ndof * sizeof(double), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return GMPNP_OK;
}

int download_vec(gmpnp_solver* s, const double* dev, double* file_order) {
  const int nf = s->nf, nv = s->t.nv;
  HIP_TRY(hipMemcpyAsync(s->h_stage, dev, (size_t)s->ndof * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  for (int i = 0; i < nv; ++i) std::memcpy(&file_order[(size_t)s->t.perm[i] * nf], &s->h_stage[(size_t)i * nf], nf * sizeof(double));
  return GMPNP_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

#ifndef GMPNP_BUILD_ID
#define GMPNP_BUILD_ID "unstamped"
#endif
const char* gmpnp_version(void) { return "gmpnp-mi355x 0.2 (gfx950)"; }
const char* gmpnp_build_id(void) { return GMPNP_BUILD_ID; }
const char* gmpnp_last_error(void) { return g_err.c_str(); }

static int create_impl(const gmpnp_mesh_t* mesh, const gmpnp_model_t* model, const gmpnp_quadrature_t* quad,
                       const gmpnp_options_t* opts, const gmpnp_partition_t* part, gmpnp_solver** out);

int gmpnp_create(const gmpnp_mesh_t* mesh, const gmpnp_model_t* model, const gmpnp_quadrature_t* quad,
                 const gmpnp_options_t* opts, gmpnp_solver** out) {
  return create_impl(mesh, model, quad, opts, nullptr, out);
}

int gmpnp_create_partition(const gmpnp_mesh_t* mesh, const gmpnp_model_t* model, const gmpnp_quadrature_t* quad,
                           const gmpnp_options_t* opts, const gmpnp_partition_t* part, gmpnp_solver** out) {
  if (!part) return fail(GMPNP_ERR_INVALID, "partition is NULL");
  if (part->size < 1 || part->rank < 0 || part->rank >= part->size || part->n_neighbours < 0) return fail(GMPNP_ERR_INVALID, "bad partition rank / size");
  if (part->n_neighbours > 0 && (!part->neighbour_rank || !part->send_ptr || !part->recv_ptr || !part->send_vertices || !part->recv_vertices))
    return fail(GMPNP_ERR_INVALID, "partition: halo plan missing");
  return create_impl(mesh, model, quad, opts, part, out);
}

static int create_impl(const gmpnp_mesh_t* mesh, const gmpnp_model_t* model, const gmpnp_quadrature_t* quad,
                       const gmpnp_options_t* opts, const gmpnp_partition_t* part, gmpnp_solver** out) {
  if (!mesh || !quad || !out) return fail(GMPNP_ERR_INVALID, "NULL argument");
  *out = nullptr;
  int rc = check_model(model, mesh->dim); if (rc) return rc;
  const int nf = model->n_species + 1;
  if (!((mesh->dim == 3 && nf == 9) || (mesh->dim == 1 && nf == 7)))
    return fail(GMPNP_ERR_INVALID, "supported spaces: 3D with 8 species + potential, 1D with 6 species + potential");
  if (quad->nq_f < 1 || quad->nq_f > GMPNP_MAX_QUAD || quad->nq_j < 1 || quad->nq_j > GMPNP_MAX_QUAD)
    return fail(GMPNP_ERR_INVALID, "quadrature size out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(GMPNP_ERR_HIP, "no HIP device visible: libgmpnp.so has no CPU fallback");
  std::unique_ptr<gmpnp_solver> s(new gmpnp_solver);
  if (opts) s->opts = *opts;
  if (s->opts.device_id < 0 || s->opts.device_id >= ndev) return fail(GMPNP_ERR_INVALID, "device_id out of range");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  // a partitioned handle keeps to one stream (no side stream) and the group drives its launches; whether the coarse workgroups
  // may ride inside the tile launches (peer transport, gmpnp_group.h) is still the caller's choice + the residency proof below
  const bool part_fused_ok = part && !s->opts.shared_device && s->opts.launch_form != 4;
  if (part) {
    if (mesh->dim != 3) return fail(GMPNP_ERR_INVALID, "mesh partitions exist for 3D meshes");
    s->opts.shared_device = 1; s->opts.launch_form = 4;
  }
  std::string err = build_topology(*mesh, nf, s->opts.n_aggregates, s->t, part);
  if (!err.empty()) return fail(GMPNP_ERR_INVALID, err);
  Topology& t = s->t;
  s->dim = mesh->dim; s->nf = nf; s->nn = mesh->dim + 1; s->ndof = t.nv * nf; s->nb = (int)t.cols.size();
  s->ncoarse = t.nagg * nf;
  s->model = *model; s->quad = *quad;
  auto conv_facets = [&](int n, const int32_t* f, std::vector<int32_t>& dst, const char* what) -> int {
    if (n < 0 || (n > 0 && !f)) return fail(GMPNP_ERR_INVALID, std::string("bad ") + what);
    dst.resize((size_t)n * 3);
    for (size_t k = 0; k < dst.size(); ++k) {
      if (f[k] < 0 || f[k] >= t.nv) return fail(GMPNP_ERR_INVALID, std::string(what) + " references a vertex outside the mesh");
      dst[k] = t.iperm[f[k]];
    }
    return GMPNP_OK;
  };
  if (mesh->dim == 3) {
    rc = conv_facets(mesh->n_wall_facets, mesh->wall_facets, s->wall_f, "wall_facets"); if (rc) return rc;
    rc = conv_facets(mesh->n_exit_facets, mesh->exit_facets, s->exit_f, "exit_facets"); if (rc) return rc;
  }
  for (int k = 0; k < mesh->n_point_vertices; ++k) {
    const int v = mesh->point_vertices[k];
    if (v < 0 || v >= t.nv) return fail(GMPNP_ERR_INVALID, "point_vertices references a vertex outside the mesh");
    s->point_v.push_back(t.iperm[v]);
  }
  HIP_TRY(hipStreamCreate(&s->stream));
  for (auto& e : s->ev_phase) HIP_TRY(hipEventCreate(&e));
  if (s->opts.krylov_batch < 0 || s->opts.profile_every < 0) return fail(GMPNP_ERR_INVALID, "negative option");

  const int nv = t.nv, nc = t.nc, nn = s->nn, ndof = s->ndof;
  const int ej_stride = (mesh->dim == 3) ? Lay<3, 9>::EJ_STRIDE : Lay<1, 7>::EJ_STRIDE;
  s->n_resblocks = grid_for(ndof, kVecBlock);
  std::vector<gmpnp_model_t> mv(1, s->model); std::vector<gmpnp_quadrature_t> qv(1, s->quad);
  HIP_TRY(s->d_model.upload(mv)); HIP_TRY(s->d_quad.upload(qv));
  HIP_TRY(s->coords.upload(t.coords)); HIP_TRY(s->cells.upload(t.cells));
  HIP_TRY(s->u.alloc(ndof)); HIP_TRY(s->un.alloc(ndof)); HIP_TRY(s->F.alloc(ndof));
  s->h_bcflag.assign(ndof, 0); s->h_bcflag_dev.assign(ndof, 0);
  HIP_TRY(hipHostMalloc((void**)&s->h_bcval, (size_t)ndof * sizeof(double)));
  HIP_TRY(hipHostMalloc((void**)&s->h_stage, (size_t)ndof * sizeof(double)));
  std::memset(s->h_bcval, 0, (size_t)ndof * sizeof(double));
  HIP_TRY(s->bcflag.upload(s->h_bcflag)); HIP_TRY(s->bcval.alloc(ndof));
  HIP_TRY(s->EF.alloc((size_t)nc * nn * nf)); HIP_TRY(s->EJ.alloc((size_t)nc * ej_stride));
  HIP_TRY(s->n2e_ptr.upload(t.n2e_ptr)); HIP_TRY(s->n2e.upload(t.n2e));
  HIP_TRY(s->rowptr.upload(t.rowptr)); HIP_TRY(s->cols.upload(t.cols));
  HIP_TRY(s->cptr.upload(t.cptr)); HIP_TRY(s->contrib.upload(t.contrib));
  { const size_t pad = (size_t)kRowPad * nf * kWave;  // the Krylov kernels preload unconditionally past short slices
    HIP_TRY(s->vals.alloc((size_t)t.slice_off[t.nslices], true, pad)); HIP_TRY(s->vals_s.alloc((size_t)t.slice_off[t.nslices], true, pad)); }
  HIP_TRY(s->slice_off.upload(t.slice_off)); HIP_TRY(s->slice_colbase.upload(t.slice_colbase));
  HIP_TRY(s->slice_node0.upload(t.slice_node0)); HIP_TRY(s->slice_nn.upload(t.slice_nn)); HIP_TRY(s->node_slice.upload(t.node_slice));
  HIP_TRY(s->sell_cols.upload(t.sell_cols)); HIP_TRY(s->sell_aggslot.upload(t.sell_aggslot));
  HIP_TRY(s->wl_slice.upload(t.wl_slice)); HIP_TRY(s->wl_kpos.upload(t.wl_kpos));
  HIP_TRY(s->sell_blk.upload(t.sell_blk));
  HIP_TRY(s->tile_slice0.upload(t.tile_slice0)); HIP_TRY(s->tile_agg.upload(t.tile_agg)); HIP_TRY(s->tile_slot.upload(t.tile_slot));
  HIP_TRY(s->tile_aggs.upload(t.tile_aggs)); HIP_TRY(s->tile_nagg.upload(t.tile_nagg));
  HIP_TRY(s->tile_rec.upload(t.tile_rec)); HIP_TRY(s->tile_cols.upload(t.tile_cols));
  HIP_TRY(s->tile_colslot.upload(t.tile_colslot)); HIP_TRY(s->sell_lcol.upload(t.sell_lcol));
  HIP_TRY(s->Dinv.alloc((size_t)nv * nf * nf));
  HIP_TRY(s->agg.upload(t.agg)); HIP_TRY(s->agg_start.upload(t.agg_start)); HIP_TRY(s->row_aggs.upload(t.row_aggs));
  HIP_TRY(s->AP.alloc((size_t)ndof * kMaxRowAggs * nf));
  // a few dozen nodes per chunk: 32 chunks per aggregate on the reference meshes, up to 1024 on refined ones (one workgroup each;
  // with the fixed 32 a twice-refined mesh spent 1.9 ms here, 770 nodes in a serial loop per thread)
  s->c.coarse_chunks = std::min(1024, std::max(kCoarseChunks, nv / std::max(1, s->t.nagg) / 24));
  HIP_TRY(s->AcPart.alloc((size_t)s->c.coarse_chunks * s->ncoarse * s->ncoarse));
  HIP_TRY(s->Ac.alloc((size_t)s->ncoarse * s->ncoarse)); HIP_TRY(s->Aci.alloc((size_t)s->ncoarse * s->ncoarse));
  HIP_TRY(s->Aci2.alloc((size_t)s->ncoarse * s->ncoarse));
  const gmpnp_options_t& po = s->opts;
  if (po.launch_form != 0 && po.launch_form != 2 && po.launch_form != 4) return fail(GMPNP_ERR_INVALID, "launch_form must be 0, 2 or 4");
  if (po.coarse_refresh < 0 || po.burst_iterations < 0 || po.warm_start < -1 || po.warm_start > 1 || !(po.band_lu_max_gb >= 0.0) ||
      po.vector_form < 0 || po.vector_form > 2)
    return fail(GMPNP_ERR_INVALID, "option out of range");
  s->coarse_async = (po.coarse_refresh == 0 && !po.shared_device) ? 1 : 0;
  s->coarse_lag = po.coarse_refresh > 0 ? po.coarse_refresh : 3;
  s->warm_async = (po.warm_in_stream || po.shared_device) ? 0 : 1;
  s->warm_start = po.warm_start == 0 ? 2 : (po.warm_start == 1 ? 1 : 0);
  s->host_poll = po.progress_by_copy ? 0 : 1;
  s->burst_iters = std::max(1, po.burst_iterations);
  s->phase_timing = po.phase_timing != 0;
  s->direct_fallback = po.no_direct_fallback ? 0 : 1;
  s->strict_steric = po.strict_steric ? 1 : 0;
  if (po.band_lu_max_gb > 0.0) s->lu_max_gb = po.band_lu_max_gb;
  if (s->coarse_async || s->warm_async) {   // the side stream exists only when something uses it
    HIP_TRY(hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_mat, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&s->ev_chain, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_jac, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&s->ev_dots, hipEventDisableTiming));
  }
  for (DevBuf<double>* b : {&s->kr, &s->krhat, &s->kp0, &s->kp1, &s->kv0, &s->kv1, &s->ks, &s->kt, &s->ky, &s->kx, &s->kxp, &s->kb}) HIP_TRY(b->alloc(ndof));
  HIP_TRY(s->yc.alloc((size_t)kMaxCoarse * 32));  // [nagg <= 16][ncoarse] column-block products (+ development stamps)
  for (DevBuf<double>* b : {&s->cpart_v0, &s->cpart_v1, &s->cpart_t, &s->cpart_r0, &s->cpart_r1, &s->cpart_p0, &s->cpart_p1})
    HIP_TRY(b->alloc((size_t)s->ncoarse * t.tile_slots));
  HIP_TRY(s->krand.alloc(ndof));
  HIP_TRY(s->ticket.alloc(16 * 66));  // counter + 64 replicated flags, one cache line each
  HIP_TRY(s->part_a.alloc((size_t)2 * t.ntiles));  // (rhat,v) partials, then ||r||^2 partials
  HIP_TRY(s->part_b.alloc((size_t)4 * t.ntiles));
  HIP_TRY(s->part_f.alloc((size_t)3 * s->n_resblocks));
  HIP_TRY(s->scal.alloc(1)); HIP_TRY(s->status.alloc(1));
  HIP_TRY(hipHostMalloc((void**)&s->h_scal, 3 * sizeof(KrylovScalars)));
  HIP_TRY(hipHostMalloc((void**)&s->h_poll, 64, hipHostMallocCoherent | hipHostMallocMapped));
  std::memset(s->h_poll, 0, 64);
  { void* dp = nullptr; HIP_TRY(hipHostGetDevicePointer(&dp, s->h_poll, 0)); s->c.poll = (HostPoll*)dp; }
  for (auto& e : s->ev_poll) HIP_TRY(hipEventCreate(&e));
  {  // Two launches per iteration (coarse workgroups inside the tile launch, tile workgroups wait for their flags) only
     // where the runtime's occupancy figure proves the whole launch resident at once: a waiting workgroup must never
     // keep the workgroup it waits for off the machine.  Not on a device the caller shares with other work.
    hipDeviceProp_t prop{};
    HIP_TRY(hipGetDeviceProperties(&prop, s->opts.device_id));
    int occ_a = 0, occ_b = 0;
    if (mesh->dim == 3) {
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_a, k_half_a<9>, kKrylovThreads, 0));
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, k_half_b<9>, kKrylovThreads, 0));
    }
    s->resident_slots = std::min(occ_a, occ_b) * prop.multiProcessorCount;
    const bool resident = mesh->dim == 3 && (s->t.ntiles + s->t.nagg) <= s->resident_slots;
    if (po.launch_form == 2 && (!resident || po.shared_device)) {
      char buf[200];
      snprintf(buf, sizeof buf, "launch_form 2 refused: %d workgroups per launch, %d resident at once (%d per CU x %d CUs)%s",
               s->t.ntiles + s->t.nagg, s->resident_slots, std::min(occ_a, occ_b), prop.multiProcessorCount,
               po.shared_device ? ", device declared shared" : "");
      return fail(GMPNP_ERR_INVALID, buf);
    }
    s->fused_half = po.launch_form == 4 ? false : (resident && !po.shared_device && !s->prereduce);
    if (part) s->fused_half = resident && part_fused_ok;
    // Vector form.  On-the-fly (the tile kernels recompute p / s at their column nodes from four / two vectors: one launch
    // per half-iteration) wins while the operands are cache resident; materialised (k_vec_a / k_vec_b write p / s for all
    // rows first, the tile kernels stage one vector) wins once the gathers cost memory bandwidth: matrix above 768 MB.
    const double matrix_mb = (double)s->nb * nf * nf * sizeof(double) / 1e6;
    s->matp = !part && mesh->dim == 3 && (po.vector_form == 1 || (po.vector_form == 0 && matrix_mb > 768.0));
    if (s->matp) {
      if (po.launch_form == 2) return fail(GMPNP_ERR_INVALID, "vector_form 1 (materialised) has its own launches: not with launch_form 2");
      s->fused_half = false;
    }
  }
  HIP_TRY(hipHostMalloc((void**)&s->h_part, 3 * std::max(s->n_resblocks, 1) * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
  HIP_TRY(hipHostMalloc((void**)&s->h_status, 64, hipHostMallocCoherent | hipHostMallocMapped));
  *s->h_status = 0;

  Ctx& c = s->c;
  c.nv = nv; c.nc = nc; c.ndof = ndof; c.nb = s->nb; c.nslices = t.nslices; c.ntiles = t.ntiles; c.n_work = (int)t.wl_slice.size();
  c.nagg = t.nagg; c.ncoarse = s->ncoarse; c.tile_slots = t.tile_slots; c.n_robin = 0; c.use_coarse = 1;
  c.model = s->d_model.p; c.quad = s->d_quad.p; c.coords = s->coords.p; c.cells = s->cells.p;
  c.u = s->u.p; c.un = s->un.p; c.F = s->F.p; c.bcflag = s->bcflag.p; c.bcval = s->bcval.p;
  c.EF = s->EF.p; c.EJ = s->EJ.p; c.n2e_ptr = s->n2e_ptr.p; c.n2e = s->n2e.p;
  c.rowptr = s->rowptr.p; c.cols = s->cols.p; c.cptr = s->cptr.p; c.contrib = s->contrib.p;
  c.vals = s->vals.p; c.vals_s = s->vals_s.p; c.slice_off = s->slice_off.p; c.slice_colbase = s->slice_colbase.p;
  c.slice_node0 = s->slice_node0.p; c.slice_nn = s->slice_nn.p; c.node_slice = s->node_slice.p;
  c.sell_cols = s->sell_cols.p; c.sell_aggslot = s->sell_aggslot.p; c.wl_slice = s->wl_slice.p; c.wl_kpos = s->wl_kpos.p;
  c.sell_blk = s->sell_blk.p; c.tile_slice0 = s->tile_slice0.p; c.tile_agg = s->tile_agg.p; c.tile_slot = s->tile_slot.p;
  c.tile_aggs = s->tile_aggs.p; c.tile_nagg = s->tile_nagg.p;
  c.tile_rec = s->tile_rec.p; c.col_stride = t.col_stride; c.tile_cols = s->tile_cols.p; c.tile_colslot = s->tile_colslot.p; c.sell_lcol = s->sell_lcol.p;
  c.Dinv = s->Dinv.p; c.agg = s->agg.p; c.agg_start = s->agg_start.p; c.row_aggs = s->row_aggs.p;
  c.AP = s->AP.p; c.AcPart = s->AcPart.p; c.Ac = s->Ac.p; c.Aci = s->Aci.p;
  s->aci_buf[0] = s->Aci.p; s->aci_buf[1] = s->Aci2.p; s->aci_cur = 0;
  c.kr = s->kr.p; c.krhat = s->krhat.p; c.kp[0] = s->kp0.p; c.kp[1] = s->kp1.p; c.kv[0] = s->kv0.p; c.kv[1] = s->kv1.p;
  c.ks = s->ks.p; c.kt = s->kt.p; c.ky = s->ky.p; c.kx = s->kx.p; c.kb = s->kb.p; c.yc = s->yc.p;
  c.cpart_r[0] = s->cpart_r0.p; c.cpart_r[1] = s->cpart_r1.p; c.cpart_p[0] = s->cpart_p0.p; c.cpart_p[1] = s->cpart_p1.p;
  c.cpart_v[0] = s->cpart_v0.p; c.cpart_v[1] = s->cpart_v1.p; c.cpart_t = s->cpart_t.p;
  c.ticket = s->ticket.p; c.part_a = s->part_a.p; c.part_rr = s->part_a.p + t.ntiles; c.part_b = s->part_b.p;
  { void* dp = nullptr; HIP_TRY(hipHostGetDevicePointer(&dp, s->h_part, 0)); c.part_f = (double*)dp;
    HIP_TRY(hipHostGetDevicePointer(&dp, s->h_status, 0)); c.status_host = (int32_t*)dp; }
  c.scal = s->scal.p; c.status = s->status.p;
  c.own_node0 = t.own_node0; c.own_node1 = t.own_node1; c.own_agg0 = t.own_agg0; c.own_agg1 = t.own_agg1; c.tile0 = t.own_tile0; c.dist = 0;
  c.wl_run_blocks = t.wl_run_blocks;
  // Pre-reduced sums for the coarse kernels (red_i / red_a / red_b).  Large meshes: an aggregate has thousands of tile slots,
  // and summing them inside every coarse workgroup (and inside every workgroup of k_minv_apply) costs more than one
  // k_dist_reduce launch per half-iteration (refine 2, before: k_coarse_a 92 us, k_minv_apply 5.5 ms per call).  Partitioned handles
  // use the same buffers for their all-reduced sums.
  HIP_TRY(s->red_i.alloc(s->ncoarse)); HIP_TRY(s->red_a.alloc(2 + 3 * (size_t)s->ncoarse)); HIP_TRY(s->red_b.alloc(4 + (size_t)s->ncoarse));
  c.red_i = s->red_i.p; c.red_a = s->red_a.p; c.red_b = s->red_b.p;
  if (s->opts.element_stores < 0 || s->opts.element_stores > 2 || (s->opts.element_stores == 2 && mesh->dim != 3))
    return fail(GMPNP_ERR_INVALID, "element_stores: 0 (automatic), 1 (direct), 2 (staged, 3D meshes)");
  s->staged_element = mesh->dim == 3 && s->opts.element_stores != 1;   // measured faster at every size: 36.6 vs 40.5 us on L_50_R_5, 1.18 vs 2.00 ms at two refinements
  s->prereduce = !part && mesh->dim == 3 && t.tile_slots > 128;
  if (s->prereduce) c.dist = 1;
  if (part) {
    s->partitioned = true; s->part_rank = part->rank; s->part_size = part->size;
    std::vector<int32_t> sn, rn;
    s->send_ptr.assign(1, 0); s->recv_ptr.assign(1, 0);
    for (int q = 0; q < part->n_neighbours; ++q) {
      if (part->neighbour_rank[q] < 0 || part->neighbour_rank[q] >= part->size || part->neighbour_rank[q] == part->rank)
        return fail(GMPNP_ERR_INVALID, "partition: bad neighbour rank");
      s->nb_rank.push_back(part->neighbour_rank[q]);
      for (int k = part->send_ptr[q]; k < part->send_ptr[q + 1]; ++k) {
        const int v = part->send_vertices[k];
        if (v < 0 || v >= nv || !part->vertex_owned[v]) return fail(GMPNP_ERR_INVALID, "partition: send list must name owned local vertices");
        sn.push_back(t.iperm[v]);
      }
      for (int k = part->recv_ptr[q]; k < part->recv_ptr[q + 1]; ++k) {
        const int v = part->recv_vertices[k];
        if (v < 0 || v >= nv || part->vertex_owned[v]) return fail(GMPNP_ERR_INVALID, "partition: receive list must name ghost vertices");
        rn.push_back(t.iperm[v]);
      }
      s->send_ptr.push_back((int32_t)sn.size()); s->recv_ptr.push_back((int32_t)rn.size());
    }
    HIP_TRY(s->send_nodes.upload(sn)); HIP_TRY(s->recv_nodes.upload(rn));
    {  // the tiles' column lists with every ghost node replaced by -(its index in the receive list + 1): exchange-prologue launches
      std::vector<int32_t> slot_of(nv, -1), tcx = t.tile_cols;
      for (size_t k = 0; k < rn.size(); ++k) slot_of[rn[k]] = (int32_t)k;
      for (auto& node : tcx) if (node >= 0 && node < nv && slot_of[node] >= 0) node = -(slot_of[node] + 1);
      HIP_TRY(s->tile_cols_x.upload(tcx));
    }
    const size_t wmax = (size_t)nf * nf;   // widest exchange: the inverse diagonal blocks of the ghost nodes
    HIP_TRY(s->sendbuf.alloc(std::max<size_t>(1, sn.size() * wmax))); HIP_TRY(s->recvbuf.alloc(std::max<size_t>(1, rn.size() * wmax)));
    HIP_TRY(s->red_norm.alloc(8));
    HIP_TRY(hipHostMalloc((void**)&s->h_red, 8 * sizeof(double)));
    c.dist = 1;
  }
  rc = rebuild_boundary(s.get()); if (rc) return rc;
  rc = build_tridiagonal(s.get()); if (rc) return rc;
  // the coarse inverse keeps its whole matrix in LDS: opt in to > 64 KiB of dynamic LDS
  if (nf == 9) HIP_TRY(hipFuncSetAttribute((const void*)k_coarse_invert<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)coarse_lds_bytes(s->ncoarse, 9)));
  else HIP_TRY(hipFuncSetAttribute((const void*)k_coarse_invert<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)coarse_lds_bytes(s->ncoarse, 7)));
  HIP_TRY(hipDeviceSynchronize());
  *out = s.release();
  return GMPNP_OK;
}

void gmpnp_destroy(gmpnp_solver* s) {
  if (!s) return;
  (void)hipSetDevice(s->opts.device_id);
  (void)hipDeviceSynchronize();
  delete s;
}

int gmpnp_set_model(gmpnp_solver* s, const gmpnp_model_t* model) {
  if (!s) return fail(GMPNP_ERR_INVALID, "NULL handle");
  int rc = check_model(model, s->dim); if (rc) return rc;
  if (model->n_species + 1 != s->nf) return fail(GMPNP_ERR_INVALID, "n_species cannot change after create");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->model = *model;
  HIP_TRY(hipMemcpy(s->d_model.p, &s->model, sizeof(gmpnp_model_t), hipMemcpyHostToDevice));
  s->jacobian_valid = false; s->precond_valid = false;
  return rebuild_boundary(s);
}

int gmpnp_set_supg(gmpnp_solver* s, const double* rho, const int32_t* w_index) {
  if (!s) return fail(GMPNP_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->jacobian_valid = false;
  if (!rho) { s->c.supg_rho = nullptr; return GMPNP_OK; }
  if (s->dim != 1) return fail(GMPNP_ERR_INVALID, "SUPG stabilisation exists for 1D meshes only (reference 1D:597-722)");
  const int ns = s->nf - 1, nv = s->t.nv;
  std::vector<double> tmp((size_t)nv * ns);
  for (int I = 0; I < nv; ++I)
    for (int j = 0; j < ns; ++j) {
      const double v = rho[(size_t)s->t.perm[I] * ns + j];
      if (!(v == v) || v < 0.0) return fail(GMPNP_ERR_INVALID, "SUPG parameters must be finite and non-negative");
      tmp[(size_t)I * ns + j] = v;
    }
  HIP_TRY(s->supg_rho.upload(tmp));
  s->c.supg_rho = s->supg_rho.p;
  for (int j = 0; j < GMPNP_MAX_SPECIES; ++j) {
    const int w = (w_index && j < ns) ? w_index[j] : j;
    if (j < ns && (w < 0 || w >= ns)) return fail(GMPNP_ERR_INVALID, "SUPG gradient index out of range");
    s->c.supg_w[j] = w;
  }
  return GMPNP_OK;
}

int gmpnp_set_dirichlet(gmpnp_solver* s, int64_t n, const int64_t* dofs, const double* values) {
  if (!s || n < 0 || (n > 0 && (!dofs || !values))) return fail(GMPNP_ERR_INVALID, "bad arguments");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  HIP_TRY(hipStreamSynchronize(s->stream));   // the pinned value buffer may still be in flight
  std::fill(s->h_bcflag.begin(), s->h_bcflag.end(), 0);
  std::memset(s->h_bcval, 0, (size_t)s->ndof * sizeof(double));
  for (int64_t k = 0; k < n; ++k) {
    const int64_t d = dofs[k];
    if (d < 0 || d >= s->ndof) return fail(GMPNP_ERR_INVALID, "Dirichlet dof out of range");
    const int r = s->t.iperm[d / s->nf] * s->nf + (int)(d % s->nf);
    s->h_bcflag[r] = 1; s->h_bcval[r] = values[k];
  }
  if (s->h_bcflag != s->h_bcflag_dev) {  // the dof SET rarely changes (3D: only the CO2 value moves between time steps)
    HIP_TRY(hipMemcpy(s->bcflag.p, s->h_bcflag.data(), s->ndof, hipMemcpyHostToDevice));
    s->h_bcflag_dev = s->h_bcflag;
  }
  HIP_TRY(hipMemcpyAsync(s->bcval.p, s->h_bcval, s->ndof * sizeof(double), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->jacobian_valid = false;
  return GMPNP_OK;
}

int gmpnp_set_state(gmpnp_solver* s, const double* u, const double* u_n) {
  if (!s) return fail(GMPNP_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (u) { int rc = upload_vec(s, u, s->u.p); if (rc) return rc; s->state_jumped = true; }
  if (u_n) { int rc = upload_vec(s, u_n, s->un.p); if (rc) return rc; }
  s->jacobian_valid = false;
  return GMPNP_OK;
}

int gmpnp_get_state(gmpnp_solver* s, double* u_out, double* u_n_out) {
  if (!s) return fail(GMPNP_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  if (u_out) { int rc = download_vec(s, s->u.p, u_out); if (rc) return rc; }
  if (u_n_out) { int rc = download_vec(s, s->un.p, u_n_out); if (rc) return rc; }
  return GMPNP_OK;
}

int gmpnp_assign_previous(gmpnp_solver* s) {
  if (!s) return fail(GMPNP_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  // stream-ordered: whatever reads u_n next is launched behind this copy, and every read-back synchronises the stream
  HIP_TRY(hipMemcpyAsync(s->un.p, s->u.p, s->ndof * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
  return GMPNP_OK;
}

int gmpnp_newton_solve(gmpnp_solver* s, const gmpnp_newton_options_t* o, gmpnp_newton_stats_t* stats) {
  if (!s || !o) return fail(GMPNP_ERR_INVALID, "NULL argument");
  if (o->maximum_iterations < 0 || o->krylov_maximum_iterations < 1) return fail(GMPNP_ERR_INVALID, "bad iteration limits");
  if (o->linear_solver < 0 || o->linear_solver > GMPNP_LINEAR_BAND_LU) return fail(GMPNP_ERR_INVALID, "unknown linear_solver");
  if (o->linear_solver == GMPNP_LINEAR_BLOCK_TRIDIAGONAL && !s->tri_ok)
    return fail(GMPNP_ERR_INVALID, "block-tridiagonal solver needs a 1D mesh in path order");
  gmpnp_newton_stats_t local{};
  gmpnp_newton_stats_t& st = stats ? *stats : local;
  st = gmpnp_newton_stats_t{};
  HIP_TRY(hipSetDevice(s->opts.device_id));
  int rc;
  GMPNP_DISPATCH(s, rc = (newton<DIM, NF>(s, *o, st)));
  return rc;
}

int32_t gmpnp_n_fields(const gmpnp_solver* s) { return s ? s->nf : 0; }
int64_t gmpnp_n_dofs(const gmpnp_solver* s) { return s ? s->ndof : 0; }
int64_t gmpnp_n_blocks(const gmpnp_solver* s) { return s ? s->nb : 0; }
int64_t gmpnp_jacobian_nnz(const gmpnp_solver* s) { return s ? (int64_t)s->nb * s->nf * s->nf : 0; }
int32_t gmpnp_n_aggregates(const gmpnp_solver* s) { return s ? s->t.nagg : 0; }
int32_t gmpnp_krylov_launches_per_iteration(const gmpnp_solver* s) { return s ? (s->fused_half ? 2 : 4) : 0; }

int gmpnp_assemble(gmpnp_solver* s, int32_t want_jacobian, double* F_out, double* norm_out) {
  if (!s) return fail(GMPNP_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  HIP_TRY(hipMemsetAsync(s->status.p, 0, sizeof(int32_t), s->stream));
  double r = 0.0; int flags = 0; int rc;
  GMPNP_DISPATCH(s, rc = (residual<DIM, NF>(s, want_jacobian != 0, &r, &flags)));
  if (rc) return rc;
  if (want_jacobian) {
    GMPNP_DISPATCH(s, rc = (launch_jac_gather<DIM, NF>(s)));
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->jacobian_valid = true; s->precond_valid = false;
  }
  if (norm_out) *norm_out = r;
  if (F_out) { rc = download_vec(s, s->F.p, F_out); if (rc) return rc; }
  if ((flags & 1) && s->strict_steric) return fail(GMPNP_ERR_NUMERIC, status_message(flags));
  return GMPNP_OK;
}

int gmpnp_get_jacobian_csr(gmpnp_solver* s, int32_t* indptr, int32_t* indices, double* data) {
  if (!s || !indptr || !indices || !data) return fail(GMPNP_ERR_INVALID, "NULL argument");
  if (!s->jacobian_valid) return fail(GMPNP_ERR_INVALID, "no Jacobian assembled for the current state");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  const Topology& t = s->t; const int nf = s->nf;
  std::vector<double> v((size_t)t.slice_off[t.nslices]);
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpy(v.data(), s->vals.p, v.size() * sizeof(double), hipMemcpyDeviceToHost));
  int64_t pos = 0; indptr[0] = 0;
  std::vector<std::pair<int, int>> order;  // (file column node, kpos)
  for (int vf = 0; vf < t.nv; ++vf) {
    const int I = t.iperm[vf], sl = t.node_slice[I], il = I - t.slice_node0[sl];
    order.clear();
    for (int k = t.rowptr[I]; k < t.rowptr[I + 1]; ++k) order.push_back({t.perm[t.cols[k]], t.sellk[k]});
    std::sort(order.begin(), order.end());
    for (int i = 0; i < nf; ++i) {
      for (auto& pr : order)
        for (int j = 0; j < nf; ++j) {
          indices[pos] = pr.first * nf + j;
          data[pos] = v[(size_t)t.slice_off[sl] + (size_t)(pr.second * nf + j) * kWave + il * nf + i];
          ++pos;
        }
      indptr[(size_t)vf * nf + i + 1] = (int32_t)pos;
    }
  }
  return GMPNP_OK;
}

int gmpnp_spmv(gmpnp_solver* s, const double* x, double* y) {
  if (!s || !x || !y) return fail(GMPNP_ERR_INVALID, "NULL argument");
  if (!s->jacobian_valid) return fail(GMPNP_ERR_INVALID, "no Jacobian assembled for the current state");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  int rc = upload_vec(s, x, s->kx.p); if (rc) return rc;
  GMPNP_DISPATCH(s, hipLaunchKernelGGL((k_spmv_plain<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c,
                                       (const double*)s->kx.p, s->kt.p));
  HIP_TRY(hipGetLastError());
  return download_vec(s, s->kt.p, y);
}

int gmpnp_linear_solve(gmpnp_solver* s, const double* b, double* x, int32_t mode, double rtol, double atol,
                       int32_t maxit, gmpnp_linear_stats_t* stats) {
  if (!s || !b || !x) return fail(GMPNP_ERR_INVALID, "NULL argument");
  if (!s->jacobian_valid) return fail(GMPNP_ERR_INVALID, "no Jacobian assembled for the current state");
  if (mode < 0 || mode > GMPNP_LINEAR_BAND_LU) return fail(GMPNP_ERR_INVALID, "unknown linear_solver");
  if (maxit < 1) return fail(GMPNP_ERR_INVALID, "max_iterations < 1");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  HIP_TRY(hipMemsetAsync(s->status.p, 0, sizeof(int32_t), s->stream));
  int rc = upload_vec(s, b, s->kr.p); if (rc) return rc;
  double bn = 0.0;
  for (int64_t i = 0; i < s->ndof; ++i) bn += b[i] * b[i];
  bn = std::sqrt(bn);
  if (mode == GMPNP_LINEAR_BLOCK_TRIDIAGONAL) {
    if (s->dim != 1) return fail(GMPNP_ERR_INVALID, "block-tridiagonal solver needs a 1D mesh");
    rc = tri_solve<7>(s, s->kr.p); if (rc) return rc;
    rc = tri_apply<7>(s, s->kx.p, 0.0, 1.0); if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(s->h_status, s->status.p, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (*s->h_status & 14) return fail(GMPNP_ERR_LINEAR, status_message(*s->h_status));
    if (stats) { stats->iterations = 1; stats->converged = 1; stats->residual_norm = 0.0; stats->rhs_norm = bn; }
    return download_vec(s, s->kx.p, x);
  }
  if (mode == GMPNP_LINEAR_BAND_LU) {
    if (s->dim != 3) return fail(GMPNP_ERR_INVALID, "block-banded LU is the 3D direct solver (1D meshes: GMPNP_LINEAR_BLOCK_TRIDIAGONAL)");
    HIP_TRY(hipMemcpyAsync(s->kb.p, s->kr.p, s->ndof * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    gmpnp_linear_stats_t ls{};
    rc = band_solve<9>(s, bn, rtol, atol, &ls);
    ls.iterations = 1;
    if (stats) *stats = ls;
    if (rc) return rc;
    return download_vec(s, s->kx.p, x);
  }
  GMPNP_DISPATCH(s, rc = (setup_preconditioner<DIM, NF>(s, mode)));
  if (rc) return rc;
  gmpnp_linear_stats_t ls{};
  GMPNP_DISPATCH(s, rc = (krylov_verified<NF>(s, mode, bn, rtol, atol, maxit, &ls)));
  if (stats) *stats = ls;
  HIP_TRY(hipMemcpy(s->h_status, s->status.p, sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipStreamSynchronize(s->stream));
  { int rc2 = drain_spmv_events(s); if (rc2) return rc2; }
  if (*s->h_status & 14) return fail(GMPNP_ERR_LINEAR, status_message(*s->h_status));
  if (rc) return rc;
  return download_vec(s, s->kx.p, x);
}

int gmpnp_attach_coarse_level(gmpnp_solver* fine, gmpnp_solver* coarse, const int32_t* parents, double theta, int32_t sweeps) {
  if (!fine || !coarse || !parents) return fail(GMPNP_ERR_INVALID, "NULL argument");
  if (fine == coarse || coarse->ml_is_coarse) return fail(GMPNP_ERR_INVALID, "a level handle serves one finer level");
  if (fine->dim != 3 || coarse->dim != 3 || fine->nf != 9 || coarse->nf != 9) return fail(GMPNP_ERR_INVALID, "multilevel term: 3D pore problems (9 fields)");
  if (fine->partitioned || coarse->partitioned) return fail(GMPNP_ERR_INVALID, "multilevel term: unpartitioned handles");
  if (fine->opts.device_id != coarse->opts.device_id) return fail(GMPNP_ERR_INVALID, "the levels live on one device");
  if (!(theta > 0.0)) return fail(GMPNP_ERR_INVALID, "theta must be positive");
  if (sweeps < 1 || sweeps > 16) return fail(GMPNP_ERR_INVALID, "sweeps: 1 ... 16");
  const int nvf = fine->t.nv, nvc = coarse->t.nv;
  if (nvc >= nvf) return fail(GMPNP_ERR_INVALID, "the coarse level has fewer vertices");
  std::vector<int32_t> par((size_t)2 * nvf), copy(nvc, -1);
  std::vector<std::vector<int32_t>> kids(nvc);
  for (int I = 0; I < nvf; ++I) {
    const int v = fine->t.perm[I];
    const int a = parents[2 * v], b = parents[2 * v + 1];
    if (a < 0 || a >= nvc || b < 0 || b >= nvc) return fail(GMPNP_ERR_INVALID, "parent vertex out of range");
    const int Ia = coarse->t.iperm[a], Ib = coarse->t.iperm[b];
    par[2 * I] = Ia; par[2 * I + 1] = (a == b) ? -1 : Ib;
    if (a == b) {
      if (copy[Ia] >= 0) return fail(GMPNP_ERR_INVALID, "two fine vertices claim to be the copy of one coarse vertex");
      copy[Ia] = I; kids[Ia].push_back(I << 1);
    } else { kids[Ia].push_back((I << 1) | 1); kids[Ib].push_back((I << 1) | 1); }   // ascending fine index: fixed summation order
  }
  std::vector<int32_t> cptr(nvc + 1, 0), clist;
  for (int Ic = 0; Ic < nvc; ++Ic) {
    if (copy[Ic] < 0) return fail(GMPNP_ERR_INVALID, "a coarse vertex has no copy on the fine level (the meshes are not nested)");
    clist.insert(clist.end(), kids[Ic].begin(), kids[Ic].end());
    cptr[Ic + 1] = (int32_t)clist.size();
  }
  HIP_TRY(hipSetDevice(fine->opts.device_id));
  HIP_TRY(fine->ml_par.upload(par)); HIP_TRY(fine->ml_child_ptr.upload(cptr)); HIP_TRY(fine->ml_child.upload(clist)); HIP_TRY(fine->ml_copy.upload(copy));
  HIP_TRY(fine->ml_z.alloc(fine->ndof));
  HIP_TRY(coarse->ml_r.alloc(coarse->ndof)); HIP_TRY(coarse->ml_w.alloc(coarse->ndof));
  fine->ml_coarse = coarse; fine->ml_theta = theta; coarse->ml_is_coarse = true; coarse->ml_sweeps = sweeps;
  // the staged operand exists in the materialised vector form only (k_vec_a / k_vec_b write the vector, the tile kernels stage one)
  fine->matp = true; fine->fused_half = false;
  fine->precond_valid = false;
  return GMPNP_OK;
}

int gmpnp_time_kernel(gmpnp_solver* s, int32_t kernel, int32_t launches, double* avg_us) {
  if (!s || !avg_us || launches < 1) return fail(GMPNP_ERR_INVALID, "bad arguments");
  if ((kernel == 0 || kernel == 18) && !s->jacobian_valid)
    return fail(GMPNP_ERR_INVALID, "no Jacobian assembled for the current state");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  hipEvent_t a, b; HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
  int rc = GMPNP_OK;
  auto one = [&]() -> int {
    int r = GMPNP_OK;
    switch (kernel) {
      case 0: GMPNP_DISPATCH(s, hipLaunchKernelGGL((k_spmv_plain<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, (const double*)s->kx.p, s->kt.p)); break;
      case 1: GMPNP_DISPATCH(s, r = (launch_element<DIM, NF>(s, true))); break;
      case 2: GMPNP_DISPATCH(s, r = (launch_jac_gather<DIM, NF>(s))); break;
      case 3: GMPNP_DISPATCH(s, r = (launch_res_gather<DIM, NF>(s))); break;
      case 4: GMPNP_DISPATCH(s, hipLaunchKernelGGL((k_bicg_a<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, 1)); break;
      case 5: GMPNP_DISPATCH(s, hipLaunchKernelGGL((k_bicg_b<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, 1)); break;
      case 6: GMPNP_DISPATCH(s, hipLaunchKernelGGL((k_coarse_a<NF>), dim3(std::max(1, s->t.nagg)), dim3(kCoarseThreads), 0, s->stream, s->c, 1)); break;
      case 7: GMPNP_DISPATCH(s, hipLaunchKernelGGL((k_coarse_b<NF>), dim3(std::max(1, s->t.nagg)), dim3(kCoarseThreads), 0, s->stream, s->c, 1)); break;
      case 8: hipLaunchKernelGGL(k_copy2, dim3(1), dim3(64), 0, s->stream, s->yc.p, (double*)nullptr, s->cpart_t.p, 64); break;
      case 9: hipLaunchKernelGGL(k_stream_read, dim3(2048), dim3(256), 0, s->stream, (const double2*)s->vals_s.p, s->vals_s.n / 2, s->part_f.p); break;
      case 10: hipLaunchKernelGGL(k_stream_read, dim3(512), dim3(256), 0, s->stream, (const double2*)s->vals_s.p, s->vals_s.n / 2, s->part_f.p); break;
      case 11: hipLaunchKernelGGL(k_stream_read, dim3(8192), dim3(256), 0, s->stream, (const double2*)s->vals_s.p, s->vals_s.n / 2, s->part_f.p); break;
      case 12: GMPNP_DISPATCH(s, hipLaunchKernelGGL((k_half_a<NF>), dim3(s->t.nagg + s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, 1,
                                                    (unsigned)(++s->fused_seq))); break;
      case 13: GMPNP_DISPATCH(s, hipLaunchKernelGGL((k_half_b<NF>), dim3(s->t.nagg + s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, 1,
                                                    (unsigned)(++s->fused_seq))); break;
      case 14: GMPNP_DISPATCH(s, hipLaunchKernelGGL((k_bicg_a_mat<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, 1)); break;
      case 15: GMPNP_DISPATCH(s, hipLaunchKernelGGL((k_bicg_b_mat<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, 1)); break;
      case 16: hipLaunchKernelGGL(k_vec_a, dim3(grid_for(s->ndof, 256)), dim3(256), 0, s->stream, s->c, 1); break;
      case 17: hipLaunchKernelGGL(k_vec_b, dim3(grid_for(s->ndof, 256)), dim3(256), 0, s->stream, s->c, 1); break;
      case 18:   // the whole 1D direct solve: extraction + ~13 levels of block cyclic reduction down and up (k_bcr_forward / _top / _backward)
        if (s->dim != 1 || !s->tri_ok) return fail(GMPNP_ERR_INVALID, "kernel 18 is the 1D block-cyclic-reduction solve");
        r = tri_solve<7>(s, s->F.p); break;
      default: return fail(GMPNP_ERR_INVALID, "unknown kernel id");
    }
    return r;
  };
  if (kernel == 12 || kernel == 13) { HIP_TRY(hipMemset(s->ticket.p, 0, 16 * 66 * sizeof(uint32_t))); s->fused_seq = 0; }
  if ((kernel >= 4 && kernel <= 7) || (kernel >= 12 && kernel <= 17)) {  // Krylov kernels: a live (not finished) solve state
    KrylovScalars z{}; z.rho[0] = z.rho[1] = 1.0; z.alpha = 1.0; z.omega = 1.0; z.beta = 0.5; z.tol = 0.0; z.max_iters = 1 << 30;
    HIP_TRY(hipMemcpy(s->scal.p, &z, sizeof z, hipMemcpyHostToDevice));
  }
  rc = one(); if (rc) return rc;  // warm
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipEventRecord(a, s->stream));
  for (int i = 0; i < launches && rc == GMPNP_OK; ++i) rc = one();
  HIP_TRY(hipEventRecord(b, s->stream));
  HIP_TRY(hipEventSynchronize(b));
  float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  if (rc) return rc;
  *avg_us = 1000.0 * ms / launches;
  if (kernel == 1 || kernel == 2) s->jacobian_valid = (kernel == 2) ? s->jacobian_valid : s->jacobian_valid;
  return GMPNP_OK;
}

int gmpnp_event_overhead(gmpnp_solver* s, int32_t pairs, double* mean_us) {
  if (!s || !mean_us || pairs < 1) return fail(GMPNP_ERR_INVALID, "bad arguments");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  hipEvent_t a, b; HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
  double sum = 0.0;
  for (int i = 0; i < pairs; ++i) {
    // keep the stream busy in front of the pair, as inside a solve
    hipLaunchKernelGGL(k_copy2, dim3(1), dim3(64), 0, s->stream, s->yc.p, (double*)nullptr, s->cpart_t.p, 64);
    HIP_TRY(hipEventRecord(a, s->stream)); HIP_TRY(hipEventRecord(b, s->stream));
    hipLaunchKernelGGL(k_copy2, dim3(1), dim3(64), 0, s->stream, s->yc.p, (double*)nullptr, s->cpart_t.p, 64);
    HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, a, b)); sum += 1000.0 * ms;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  *mean_us = sum / pairs;
  return GMPNP_OK;
}

#ifdef GMPNP_DEV_HOOKS  // development builds only (tools/): not part of the ABI, not in the shipped library
// resident workgroups per CU of the fused 3D Krylov kernels
int gmpnp_debug_occupancy(int* out4) {
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out4[0], k_bicg_a<9>, kKrylovThreads, 0));
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out4[1], k_bicg_b<9>, kKrylovThreads, 0));
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out4[2], k_spmv_plain<9>, kKrylovThreads, 0));
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out4[3], k_coarse_a<9>, kCoarseThreads, 0));
  return GMPNP_OK;
}

// debug only (not declared in gmpnp.h): raw device buffers, internal order
int gmpnp_debug_read(gmpnp_solver* s, int which, double* out, int64_t n) {
  if (!s || !out) return fail(GMPNP_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  const double* src = nullptr; int64_t cap = 0;
  switch (which) {
    case 0: src = s->krhat.p; cap = s->ndof; break;
    case 1: src = s->vals_s.p; cap = (int64_t)s->vals_s.n; break;
    case 2: src = s->Dinv.p; cap = (int64_t)s->Dinv.n; break;
    case 3: src = s->c.Aci; cap = (int64_t)s->Aci.n; break;
    case 4: src = s->vals.p; cap = (int64_t)s->vals.n; break;
    case 5: src = s->Ac.p; cap = (int64_t)s->Ac.n; break;
    case 6: src = s->kr.p; cap = s->ndof; break;
    case 7: src = s->ky.p; cap = s->ndof; break;
    case 8: src = s->ks.p; cap = s->ndof; break;
    case 9: src = s->kt.p; cap = s->ndof; break;
    case 10: src = s->kp0.p; cap = s->ndof; break;
    case 11: src = s->kp1.p; cap = s->ndof; break;
    case 12: src = s->kv0.p; cap = s->ndof; break;
    case 13: src = s->kv1.p; cap = s->ndof; break;
    case 14: src = s->yc.p; cap = (int64_t)s->yc.n; break;
    default: return GMPNP_ERR_INVALID;
  }
  if (n > cap) n = cap;
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpy(out, src, n * sizeof(double), hipMemcpyDeviceToHost));
  return (int)0;
}
#endif  // GMPNP_DEV_HOOKS

int gmpnp_spmv_profile(gmpnp_solver* s, int64_t* n_sampled, double* mean_us, int64_t* n_launched) {
  if (!s) return fail(GMPNP_ERR_INVALID, "NULL handle");
  HIP_TRY(hipSetDevice(s->opts.device_id));
  HIP_TRY(hipStreamSynchronize(s->stream));
  { int rc = drain_spmv_events(s); if (rc) return rc; }
  if (n_sampled) *n_sampled = s->spmv_sampled;
  if (mean_us) *mean_us = s->spmv_sampled ? s->spmv_us_sum / s->spmv_sampled : 0.0;
  if (n_launched) *n_launched = s->spmv_launched;
  s->spmv_sampled = 0; s->spmv_us_sum = 0.0; s->spmv_launched = 0;
  return GMPNP_OK;
}

}  // extern "C"

#include "gmpnp_group.h"
#include "gmpnp_project.h"
