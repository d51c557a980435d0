/* gmpnp.h — C ABI of libgmpnp.so: MI355X (gfx950) backend for the GMPNP time-stepping Newton solve.
 *
 * Drop-in boundary.  The reference has no native code and no FFI; the operator API of its
 * hot path is the FEniCS call
 *     solve(F == 0, u, bcs, solver_parameters={'nonlinear_solver':'newton','newton_solver':{...}})
 * at 3D/MPNP_CO2ER_pore.py:789-799 and 1D/MPNP_CO2ER_EDL.py:717,729,737, fed by the objects built at
 * 3D:329-332 (Mesh), 3D:368-382 (boundary markers / ds), 3D:404-409 (MixedElement([P1]*9)),
 * 3D:425-432 (u, u_n), 3D:460-467 (DirichletBC list), 3D:474-769 (forms), 3D:856 (u_n.assign(u)) and
 * their 1D counterparts (1D:231-234, 300-306, 320-326, 350-355, 383-595, 796).  Each entry point below
 * names the reference construct it replaces.  A reference-side binding is a ctypes stub
 * (INTEGRATION.md).
 *
 * Conventions: plain C types; caller-allocated contiguous fp64 / int32 / int64 host buffers; all
 * vertex-indexed data in mesh-FILE vertex order, dof = vertex * n_fields + field (fields = species in
 * mixed-space order, potential last); the library owns all device memory behind the opaque handle;
 * every function returns 0 or a negative gmpnp_status and never throws; one host thread per handle; a handle
 * owns its HIP streams (the main one and a side stream for work that is off the critical path of a Newton
 * iteration, ordered by events) and a few pinned host cache lines the kernels report progress into; calls
 * block until their result is on the host (gmpnp_assign_previous is stream-ordered: it returns at once, and
 * whatever reads u_n next is queued behind the copy).
 */
#ifndef GMPNP_H
#define GMPNP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMPNP_MAX_SPECIES 8
#define GMPNP_MAX_BILINEAR 4
#define GMPNP_MAX_QUAD 16
#define GMPNP_MAX_NEWTON_HISTORY 64

typedef enum {
  GMPNP_OK = 0,
  GMPNP_ERR_INVALID = -1,        /* bad argument / unsupported configuration */
  GMPNP_ERR_HIP = -2,            /* HIP runtime error (see gmpnp_last_error)   */
  GMPNP_ERR_NOT_CONVERGED = -3,  /* Newton hit maximum_iterations ([3P] error_on_nonconvergence) */
  GMPNP_ERR_LINEAR = -4,         /* Krylov breakdown / did not reach tolerance / singular block */
  GMPNP_ERR_NUMERIC = -5         /* NaN residual or 1 - sum_j a_j u_j <= 0 at a quadrature point */
} gmpnp_status;

/* Coefficient tables of one member of the GMPNP weak-form family (replaces the UFL forms
 * 3D:505-769 / 1D:383-595 and the Constants feeding them 3D:261-324,474-499 / 1D:178-208,371-375).
 *   -R_i(u) = rc0_i + sum_j rc1_ij u_j + sum_t rc2_it u_{bil_j[t]} u_{bil_k[t]}
 *   eps(u)  = eps0 + sum_j epsc_j u_j ;  charge term  sum_j qzb_j u_j  (qzb_j = q z_j bulk_j)        */
typedef struct {
  int32_t dim;        /* 1 (interval) or 3 (tetrahedron) */
  int32_t n_species;  /* 6 (1D) or 8 (3D); n_fields = n_species + 1 */
  int32_t n_bilinear;
  int32_t steric;     /* 1 = MPNP (u_i/(1-S) term), 0 = PNP */
  double inv_dt;      /* 1/del_t (3D:534) or 1/(del_t*L_D) (1D:458) */
  double q;
  double eps0;
  double z[GMPNP_MAX_SPECIES];
  double a[GMPNP_MAX_SPECIES];    /* scale_vol */
  double qzb[GMPNP_MAX_SPECIES];
  double epsc[GMPNP_MAX_SPECIES];
  double rc0[GMPNP_MAX_SPECIES];
  double rc1[GMPNP_MAX_SPECIES][GMPNP_MAX_SPECIES];
  double rc2[GMPNP_MAX_SPECIES][GMPNP_MAX_BILINEAR];
  int32_t bil_j[GMPNP_MAX_BILINEAR];
  int32_t bil_k[GMPNP_MAX_BILINEAR];
  double wall_flux[GMPNP_MAX_SPECIES];   /* J_X_wall on ds(2)            3D:474-481 */
  double exit_kappa[GMPNP_MAX_SPECIES];  /* kappa_X (u_X - 1) on ds(3)   3D:484-499 */
  double point_flux[GMPNP_MAX_SPECIES];  /* J_X at the OHP vertex        1D:371-375,553,738 */
} gmpnp_model_t;

/* Quadrature of the rational steric integrand: barycentric points (first dim+1 entries of each row
 * used) and weights summing to 1, for the residual (degree 3) and the Jacobian (degree 4) — replaces the
 * FFC/FIAT-generated rules ([3P]). */
typedef struct {
  int32_t nq_f, nq_j;
  double lam_f[GMPNP_MAX_QUAD][4];
  double w_f[GMPNP_MAX_QUAD];
  double lam_j[GMPNP_MAX_QUAD][4];
  double w_j[GMPNP_MAX_QUAD];
} gmpnp_quadrature_t;

/* Mesh + boundary facet sets — replaces Mesh(...) 3D:329-332 / 1D:231-234, the marked ds measure
 * 3D:368-382 and FunctionSpace(mesh, MixedElement([P1]*n)) 3D:404-408 / 1D:300-304. */
typedef struct {
  int32_t dim;
  int32_t n_vertices;
  int32_t n_cells;
  const double* coords;        /* [n_vertices][dim] */
  const int32_t* cells;        /* [n_cells][dim+1]  */
  const int32_t* perm;         /* [n_vertices] internal->file vertex order, or NULL (identity).
                                  Contiguous internal ranges become the coarse-space aggregates and
                                  the multi-GPU partitions, so pass a slab ordering (see
                                  gmpnp_amd.backend.slab_permutation). */
  int32_t n_wall_facets;
  const int32_t* wall_facets;  /* [n][3] vertex triples of the exterior facets in ds(2) */
  int32_t n_exit_facets;
  const int32_t* exit_facets;  /* [n][3] ds(3) */
  int32_t n_point_vertices;
  const int32_t* point_vertices; /* 1D: vertices that receive model.point_flux */
} gmpnp_mesh_t;

/* Linear solver for J dx = b (replaces PETSc KSP preonly + LU: 'mumps' 3D:792, default LU 1D:357-364). */
typedef enum {
  GMPNP_LINEAR_BICGSTAB_TWOLEVEL = 0, /* BiCGStab, right-preconditioned by node-block Jacobi + slab-aggregate coarse correction */
  GMPNP_LINEAR_BICGSTAB_JACOBI = 1,   /* BiCGStab, node-block Jacobi only ([3P] 'bicgstab' + 'jacobi') */
  GMPNP_LINEAR_BLOCK_TRIDIAGONAL = 2, /* direct block-tridiagonal LU (1D meshes only) */
  GMPNP_LINEAR_BAND_LU = 3            /* direct block-banded LU in slab order (3D meshes); also what a 3D Krylov solve
                                         that does not converge falls back to, MUMPS never failing in the reference */
} gmpnp_linear_kind;

/* newton_solver parameter dict of the reference (3D:789-798, 1D:357-364) + krylov_solver sub-dict. */
typedef struct {
  int32_t maximum_iterations;   /* 50 */
  double relative_tolerance;    /* 1e-4 */
  double absolute_tolerance;    /* 1e-4 */
  double relaxation_parameter;  /* 0.9 (3D), 1.0 (1D) */
  int32_t linear_solver;        /* gmpnp_linear_kind */
  double krylov_relative_tolerance; /* on ||b - A x|| / ||b||; 1e-10 ~ "exact-equivalent" */
  double krylov_absolute_tolerance;
  int32_t krylov_maximum_iterations;
} gmpnp_newton_options_t;

typedef struct {
  int32_t iterations;           /* Newton iterations performed ([3P] "Newton iteration k") */
  int32_t converged;
  int32_t krylov_iterations;    /* total over the solve */
  int32_t n_residuals;          /* iterations + 1 */
  double residuals[GMPNP_MAX_NEWTON_HISTORY]; /* ||b||_2 before iteration 0 and after each update */
  int32_t krylov_per_iteration[GMPNP_MAX_NEWTON_HISTORY];
  double ms_assemble, ms_setup, ms_krylov, ms_total; /* ms_total: host wall clock of the solve; the three phases are
                                                        device times (hipEvents), filled with gmpnp_options_t.phase_timing only */
  int32_t direct_solves;        /* Newton iterations whose system the block-banded LU solved (mode 3 or fallback) */
  int32_t steric_excursion;     /* 1 = some residual evaluation of this solve met 1 - sum_j a_j u_j <= 0 at a quadrature point
                                   (an iterate outside the admissible set).  UFL/FFC evaluate the quotient u_i/(1 - S) as it
                                   stands (3D:534-750, 1D:457-593), so by default this is information, not an error */
} gmpnp_newton_stats_t;

typedef struct {
  int32_t iterations;
  int32_t converged;
  double residual_norm;   /* recurrence ||r||_2 at exit */
  double rhs_norm;
} gmpnp_linear_stats_t;

/* Creation-time tunables (no reference counterpart). Zero-initialise for defaults: every field's 0 is the default. */
typedef struct {
  int32_t device_id;      /* HIP device ordinal */
  int32_t n_aggregates;   /* coarse-space slabs; 0 = default (8; at most 16 and what the LDS-resident coarse inverse allows) */
  int32_t shared_device;  /* 1 = other handles or processes use this GPU at the same time: no in-launch hand-over (four
                             launches per BiCGStab iteration) and no second HIP stream */
  int32_t krylov_batch;   /* iterations of the first burst of a BiCGStab solve; 0 = default (sized from the
                             previous solves; afterwards the host keeps one iteration queued ahead of the progress
                             the kernels report into pinned memory) */
  int32_t profile_every;  /* N > 0: the first burst of every Nth Krylov solve runs between a pair of HIP events (gmpnp_spmv_profile); 0 = off */
  int32_t launch_form;    /* launches per BiCGStab iteration: 0 = automatic (2 when hipOccupancyMaxActiveBlocksPerMultiprocessor
                             proves every workgroup of a launch resident at once, else 4), 2, 4.  Asking for 2 on a
                             problem that is not resident is refused (GMPNP_ERR_INVALID). */
  int32_t warm_start;     /* start of Newton iteration k+1's linear solve: 0 = second-order prediction from the two previous
                             corrections (default), 1 = first order, -1 = zero */
  int32_t coarse_refresh; /* coarse operator: 0 = rebuilt every Newton iteration on the side stream and adopted one iteration
                             later (default), N > 0 = rebuilt in the main stream every Nth Newton iteration */
  int32_t progress_by_copy; /* 1 = the host polls BiCGStab with a device-to-host copy + event per burst instead of the
                               pinned progress mirror */
  int32_t burst_iterations; /* iterations queued per poll after the first burst; 0 = 1 */
  int32_t phase_timing;   /* 1 = fill ms_assemble / ms_setup / ms_krylov of the Newton statistics (five event records
                             and one wait more per iteration) */
  int32_t no_direct_fallback; /* 1 = a 3D Krylov solve that does not converge is an error instead of a block-banded LU solve */
  int32_t warm_in_stream; /* 1 = the test of the predicted start runs in the main stream behind the set-up */
  int32_t vector_form;    /* BiCGStab half-iterations: 2 = the tile kernels recompute p / s at their column nodes on the fly (one launch
                             per half-iteration), 1 = streaming kernels write p / s for all rows first and the tile kernels stage
                             one vector (wins when the gathers cost memory bandwidth), 0 = automatic (1 above 768 MB of matrix) */
  int32_t strict_steric;  /* 1 = an iterate with 1 - sum_j a_j u_j <= 0 at a quadrature point ends the solve with GMPNP_ERR_NUMERIC
                             (rounds 1-2).  0 (default) = the reference's behaviour: no such test; an iterate that overshoots
                             and comes back converges, one that does not ends as NaN / not converged */
  int32_t element_stores; /* how the element kernel writes its per-cell records: 0 = automatic (3D: staged through LDS and written
                             record by record, up to 512 contiguous bytes per store instruction; 1D: direct), 1 = direct stores of
                             one lane per cell (every store instruction touches 64 records 1.6 KB apart), 2 = staged */
  double band_lu_max_gb;  /* largest band storage the direct solver may allocate; 0 = 48 */
} gmpnp_options_t;

typedef struct gmpnp_solver gmpnp_solver;

const char* gmpnp_version(void);
/* First 16 hex digits of the SHA-256 over the library's native sources (stamped by the build recipe, __graft_entry__.build):
 * measurements kept in the repository (PMC traffic per launch) name the build they were taken on. */
const char* gmpnp_build_id(void);
/* Message of the most recent failure on this thread. */
const char* gmpnp_last_error(void);

/* Mesh + FunctionSpace + forms -> device-resident problem. u and u_n start as zeros ([3P] Function(V), 3D:425). */
int gmpnp_create(const gmpnp_mesh_t* mesh, const gmpnp_model_t* model, const gmpnp_quadrature_t* quad,
                 const gmpnp_options_t* opts, gmpnp_solver** out);
void gmpnp_destroy(gmpnp_solver* s);

/* Re-upload constants (Constant(...) objects rebuilt between steps: del_t 1D:639-642 (Q2), J_OH/J_H 1D:789-793). */
int gmpnp_set_model(gmpnp_solver* s, const gmpnp_model_t* model);

/* SUPG stabilisation of the PNP model, reference 1D/MPNP_CO2ER_EDL.py:597-722 (--model PNP --stabilization Y; 1D
 * meshes only): F gets  - sum_i rho_i z_i [ (u_i - u_i^n)/(dt L_D) + z_i grad(w_i).grad(p) + R_i ] grad(p).grad(v_i) dx
 * with the P1 field rho_i given by its vertex values rho[vertex*n_species + i] (0 = species not stabilised; the
 * reference recomputes them every time step from the previous potential, 1D:650-685) and w_i = u_{w_index[i]} (NULL =
 * identity; the reference's OH term takes grad(u_H)). rho = NULL switches the terms off. */
int gmpnp_set_supg(gmpnp_solver* s, const double* rho, const int32_t* w_index);

/* bcs list after DOLFIN's in-order application (later wins): unique dofs + values (3D:460-467,835-838; 1D:350-355). */
int gmpnp_set_dirichlet(gmpnp_solver* s, int64_t n, const int64_t* dofs, const double* values);

/* u / u_n contents (either pointer may be NULL to leave that vector untouched). interpolate 3D:432, project 1D:326. */
int gmpnp_set_state(gmpnp_solver* s, const double* u, const double* u_n);
/* compute_vertex_values() of all fields at once (3D:802-813, 1D:745-753). */
int gmpnp_get_state(gmpnp_solver* s, double* u_out, double* u_n_out);
/* u_n.assign(u) (3D:856, 1D:796), on device. */
int gmpnp_assign_previous(gmpnp_solver* s);

/* solve(F == 0, u, bcs, solver_parameters) (3D:789-799, 1D:737-742): damped Newton on the device state u.
 * Returns GMPNP_ERR_NOT_CONVERGED where DOLFIN raises RuntimeError; stats are filled either way. */
int gmpnp_newton_solve(gmpnp_solver* s, const gmpnp_newton_options_t* opts, gmpnp_newton_stats_t* stats);

/* ---- lower-level hooks for parity tests and benchmarks ([3P] assemble / DirichletBC.apply / KSP) ---- */
int32_t gmpnp_n_fields(const gmpnp_solver* s);
int64_t gmpnp_n_dofs(const gmpnp_solver* s);
int64_t gmpnp_n_blocks(const gmpnp_solver* s);   /* node blocks of the BSR Jacobian */
int64_t gmpnp_jacobian_nnz(const gmpnp_solver* s); /* n_blocks * n_fields^2 */
int32_t gmpnp_n_aggregates(const gmpnp_solver* s);
/* Kernel launches per BiCGStab iteration of the 3D solver: 4 (coarse, tile, coarse, tile) or 2 (the coarse workgroups
 * ride inside the tile launches; chosen when the occupancy query proves all workgroups of a launch resident at once; gmpnp_options_t.launch_form
 * overrides). Diagnostics for the bench; no reference counterpart. */
int32_t gmpnp_krylov_launches_per_iteration(const gmpnp_solver* s);

/* b = assemble(F) with bc rows b = x - g; optionally A = assemble(J) with identity bc rows (kept on device).
 * F_out (n_dofs) and norm_out may be NULL. */
int gmpnp_assemble(gmpnp_solver* s, int32_t want_jacobian, double* F_out, double* norm_out);
/* Current device Jacobian as CSR in file-order dof numbering, columns ascending (indptr n_dofs+1, others nnz). */
int gmpnp_get_jacobian_csr(gmpnp_solver* s, int32_t* indptr, int32_t* indices, double* data);
/* y = J x with the current device Jacobian. */
int gmpnp_spmv(gmpnp_solver* s, const double* x, double* y);
/* Solve J x = b with the current device Jacobian (preconditioner is rebuilt). */
int gmpnp_linear_solve(gmpnp_solver* s, const double* b, double* x, int32_t linear_solver, double rtol,
                       double atol, int32_t max_iterations, gmpnp_linear_stats_t* stats);

/* ---- post-processing on the device ------------------------------------------------------------
 * project(sign * grad(f), W).compute_vertex_values() of a P1 field f given by its vertex values (file order): the
 * consistent-mass L2 projection of the cell-wise constant gradient the reference computes for `field_values` and the
 * `<X>_grad` arrays (1D/MPNP_CO2ER_EDL.py:802-805, 3D/MPNP_CO2ER_pore.py:884-909).  out: [n_vertices][dim], row major.
 * Jacobi-preconditioned CG on the P1 mass matrix to 1e-14 relative residual; stats (may be NULL) reports the iterations. */
int gmpnp_project_gradient(gmpnp_solver* s, const double* nodal_values, double sign, double* out, gmpnp_linear_stats_t* stats);
/* project(f, Y) of a cell-wise constant field with ncomp <= 4 components, cell_values [n_cells][ncomp] in mesh-file cell order
 * (reference 1D:599 project(CellDiameter(mesh)), 1D:651-653 the projected gradient norm of the SUPG parameters). out: [n_vertices][ncomp]. */
int gmpnp_project_cellwise(gmpnp_solver* s, int32_t ncomp, const double* cell_values, double* out, gmpnp_linear_stats_t* stats);

/* ---- mesh-partitioned solve (no reference counterpart: the reference is a serial script) -------------
 * One handle per rank on the rank's LOCAL mesh = every cell that touches an owned vertex; the other vertices of those cells
 * are ghosts (owned by a neighbouring rank), flagged as Dirichlet dofs by the caller so that their matrix rows are identity
 * rows.  Cut cells are assembled on both sides, so owned rows are complete without matrix communication.  Inside the
 * library: ghost rows of (r, v, p) / (s, t) travel after each BiCGStab half-iteration (grouped ncclSend/ncclRecv on the
 * solver's stream), the dot products and the coarse restrictions of a half-iteration travel in ONE ncclAllReduce, the
 * coarse operator is GLOBAL (slabs numbered over the whole mesh, one all-reduce of the Galerkin matrix per set-up, inverted
 * redundantly by every rank).  gmpnp_amd/dist.py builds the partition and the plan. */
typedef struct {
  int32_t rank, size;
  int32_t n_global_aggregates;       /* coarse slabs over the WHOLE mesh (<= 15 for 9 fields) */
  const int32_t* vertex_aggregate;   /* [n_vertices] slab of each LOCAL vertex (local mesh-file order); mesh.perm must run
                                        through the slabs in ascending order */
  const uint8_t* vertex_owned;       /* [n_vertices] 1 = owned by this rank, 0 = ghost; a slab is all owned or all ghost */
  int32_t n_neighbours;
  const int32_t* neighbour_rank;     /* [n_neighbours] */
  const int32_t* send_ptr;           /* [n_neighbours+1] into send_vertices */
  const int32_t* send_vertices;      /* local vertices whose rows go to neighbour q, in q's receive order */
  const int32_t* recv_ptr;           /* [n_neighbours+1] into recv_vertices */
  const int32_t* recv_vertices;      /* local ghost vertices filled from neighbour q, in q's send order */
} gmpnp_partition_t;

int gmpnp_create_partition(const gmpnp_mesh_t* local_mesh, const gmpnp_model_t* model, const gmpnp_quadrature_t* quad,
                           const gmpnp_options_t* opts, const gmpnp_partition_t* part, gmpnp_solver** out);

/* RCCL communicator of the ranks (one process per GPU).  Rank 0 makes the id and hands the 128 bytes to the others by any
 * means (the Python driver broadcasts them with torch.distributed); every rank then joins.  librccl.so is loaded on first
 * use; its absence is an error here, not at library load. */
#define GMPNP_COMM_ID_BYTES 128
typedef struct gmpnp_comm gmpnp_comm;
int gmpnp_comm_unique_id(char id[GMPNP_COMM_ID_BYTES]);
int gmpnp_comm_create(const char id[GMPNP_COMM_ID_BYTES], int32_t rank, int32_t size, int32_t device_id, gmpnp_comm** out);
/* n doubles sent to this rank itself and received back (grouped ncclSend + ncclRecv), then all-reduced over the communicator:
 * exercises every RCCL entry point of the partitioned solve; max_error = largest deviation from the expected sums. */
int gmpnp_comm_selftest(gmpnp_comm* c, int32_t n, double* max_error);
void gmpnp_comm_destroy(gmpnp_comm* c);

/* The partition handles ONE PROCESS drives: exactly one with a communicator (production: one rank per GPU), or all `size`
 * of them with comm = NULL (rehearsal of the whole partitioned algorithm inside one process on one GPU: the exchanges
 * become device copies between the handles; this is what the single-GPU test box runs). */
typedef struct gmpnp_group gmpnp_group;
int gmpnp_group_create(int32_t n_local, gmpnp_solver* const* handles, gmpnp_comm* comm, gmpnp_group** out);
/* Third transport: the caller moves the bytes.  The library stages each all-reduce / ghost exchange through pinned host
 * buffers and calls back; the Python driver implements the two calls with torch.distributed on `gloo`.  For machines
 * without RCCL between the ranks and for the two-ranks-on-one-card test (RCCL refuses two ranks on one device); every
 * collective costs two PCIe copies and a stream synchronisation, so this is not a production path.
 *   allreduce(user, buf, n):  sum buf[0..n) over all ranks, in place, same result on every rank; 0 = ok
 *   exchange(user, n_neighbours, neighbour_rank, send_offset, send_count, send_buf, recv_offset, recv_count, recv_buf):
 *       send send_buf[send_offset[j] .. +send_count[j]) (doubles) to neighbour j, receive its message into
 *       recv_buf[recv_offset[j] .. +recv_count[j]); 0 = ok */
typedef struct {
  int32_t rank, size;
  int (*allreduce)(void* user, double* buf, int32_t n);
  int (*exchange)(void* user, int32_t n_neighbours, const int32_t* neighbour_rank, const int64_t* send_offset, const int64_t* send_count,
                  const double* send_buf, const int64_t* recv_offset, const int64_t* recv_count, double* recv_buf);
  void* user;
} gmpnp_host_transport_t;
int gmpnp_group_create_hosted(gmpnp_solver* handle, const gmpnp_host_transport_t* transport, gmpnp_group** out);
/* Fourth transport: peer mailboxes — the one to use with one process per GPU.  Every rank owns a mailbox in its GPU's memory
 * (uncached), mapped into the other ranks' processes through an IPC handle; a collective is ONE kernel launch per rank that
 * stores its contribution / its ghost rows straight into the other ranks' mailboxes (xGMI between GPUs), raises a flag there
 * and waits for the flags in its own: no collective library and no host step between two BiCGStab half-iterations.
 *   1. every rank: gmpnp_group_peer_begin(handle, &group, my_handle)         -> 64 bytes to publish
 *   2. the caller gathers the handles of all ranks in rank order (any channel; this gather is also the point after which
 *      every mailbox exists), then every rank: gmpnp_group_peer_connect(group, all_handles)
 *   3. gmpnp_group_newton_solve ... ; before gmpnp_group_destroy the caller makes sure (a barrier of its own) that no rank is
 *      still inside a solve.
 * A rank that does not arrive within 5 s ends the others' launch with an error instead of a hang. */
#define GMPNP_PEER_HANDLE_BYTES 64
int gmpnp_group_peer_begin(gmpnp_solver* handle, gmpnp_group** out, char ipc_handle[GMPNP_PEER_HANDLE_BYTES]);
int gmpnp_group_peer_connect(gmpnp_group* g, const char* all_handles);
void gmpnp_group_destroy(gmpnp_group* g);
/* solve(F == 0, u, bcs, solver_parameters) on the partitioned state (each handle's u / u_n hold owned + ghost values, set
 * with gmpnp_set_state; ghost values of u are kept current inside).  Collective: every rank calls it.  Statistics are
 * identical on all ranks.  Linear solver: GMPNP_LINEAR_BICGSTAB_TWOLEVEL or _JACOBI. */
int gmpnp_group_newton_solve(gmpnp_group* g, const gmpnp_newton_options_t* opts, gmpnp_newton_stats_t* stats);
/* u_n.assign(u) on every local handle. */
/* One pass of each collective of the partitioned solve over the group's own transport with self-checking contents (an all-reduce
 * of 5 doubles, one ghost-row message per neighbour; on a peer-mailbox group that will run its solves with the exchange inside the
 * next launch, the same contents once more through the flagged-word areas those launches use): *max_error = largest deviation this
 * process saw (0 expected).  Collective:
 * every rank calls it.  No reference counterpart (the reference is serial); it is the start-up check of BASELINE configs[3]. */
int gmpnp_group_selftest(gmpnp_group* g, double* max_error);
/* Peer-mailbox groups: how the exchange of a BiCGStab half-iteration's sums and boundary rows is launched.  form 0 (default) = in
 * front of the NEXT half-iteration's coarse workgroups, inside that launch (two launches per iteration, as on one GPU), wherever
 * the launch with its exchange workgroups is resident at once; form 1 = its own launch between the two (four per iteration).
 * Every rank of a partition must run the same form: the caller compares gmpnp_group_exchange_form over the ranks after set-up and
 * sets form 1 everywhere unless all of them report 2 (gmpnp_amd/dist.py does).  gmpnp_group_exchange_form: what a solve
 * of this group will do — 2 = exchange inside the next launch, 1 = separate launches, 0 = not a peer-mailbox group.  No reference
 * counterpart. */
int gmpnp_group_set_exchange_form(gmpnp_group* g, int32_t form);
int32_t gmpnp_group_exchange_form(const gmpnp_group* g);
int gmpnp_group_assign_previous(gmpnp_group* g);

/* Geometric multilevel term of the preconditioner on uniformly refined meshes (no reference counterpart: the reference solves
 * with MUMPS, 3D:792; this keeps the iteration count of the Krylov stand-in from growing with the refinement level).
 * `coarse` is an ordinary handle of the PARENT mesh of `fine`'s mesh (same model; its Dirichlet set decides its identity rows);
 * parents[2 v + {0, 1}] = the two coarse vertices (coarse FILE order) fine vertex v (fine file order) interpolates from, both the
 * same vertex when v is a copy of it (red refinement: every fine vertex is one or the other).  From then on every preconditioner
 * set-up of `fine` injects the state into `coarse`, assembles the Jacobian there and sets up the coarse handle's own two-level
 * preconditioner M_c^-1, and M^-1 of `fine` gains  theta * P S_c P^T  (additive), S_c = `sweeps` Richardson sweeps x <- x +
 * PRE_c (r - J_c x) from x = 0 with PRE_c = M_c^-1 + [the same term for a level attached below `coarse`] — sweeps = 1 is purely
 * additive, every further sweep costs one SpMV with J_c (1/8 of a fine one) (csrc/gmpnp_multilevel.h).
 * Chains: attach level 2 to level 1, then level 1 to level 0.  `coarse` must outlive `fine` and must not be driven by the caller
 * any more.  BiCGStab then runs in the materialised vector form.  3D, unpartitioned handles on one device. */
int gmpnp_attach_coarse_level(gmpnp_solver* fine, gmpnp_solver* coarse, const int32_t* parents, double theta, int32_t sweeps);

/* Benchmark hooks: time `launches` back-to-back launches of one kernel on the handle's stream with HIP
 * events; kernel: 0 = plain Jacobian SpMV, 1 = element kernel (F+J), 2 = Jacobian gather, 3 = residual gather,
 * 4/5 = fused BiCGStab half-iterations A/B, 6/7 = their scalar+coarse kernels, 8 = one-wave copy, 9-11 = streaming
 * read of the matrix buffer with 2048 / 512 / 8192 workgroups (bandwidth probes), 12/13 = the two-launch form of the
 * half-iterations (coarse workgroups inside the tile launch), 14/15 = the tile kernels of the materialised vector form,
 * 16/17 = its streaming vector updates, 18 = one whole 1D direct solve (block cyclic reduction: extraction, every level down
 * and up; needs an assembled Jacobian). */
int gmpnp_time_kernel(gmpnp_solver* s, int32_t kernel, int32_t launches, double* avg_us);
/* Fused BiCGStab half-iterations (SpMV + vector updates) timed with HIP events since the last call (opts.profile_every):
 * n_sampled = half-iterations inside the timed bursts (each a run of back-to-back launches, all of them before the end of
 * their solve), mean_us = elapsed time / n_sampled (launch gaps included), n_launched = all half-iterations launched. */
int gmpnp_spmv_profile(gmpnp_solver* s, int64_t* n_sampled, double* mean_us, int64_t* n_launched);
/* Mean elapsed time of an EMPTY event pair on the handle's stream (what a sampled launch's bracket adds). */
int gmpnp_event_overhead(gmpnp_solver* s, int32_t pairs, double* mean_us);

#ifdef __cplusplus
}
#endif
#endif /* GMPNP_H */
