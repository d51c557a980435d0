// Internal declarations shared by the host-side topology builder and the HIP kernels of libgmpnp.so.
// Target: gfx950 (MI355X) only.  Nothing here is part of the public ABI (see include/gmpnp.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "gmpnp.h"

namespace gmpnp {

constexpr int kWave = 64;           // CDNA wavefront
constexpr int kVecBlock = 256;      // threads per workgroup of the vector kernels
constexpr int kMaxRowAggs = 4;      // distinct coarse aggregates a block row may touch
constexpr int kMaxCoarse = 140;     // coarse dimension limit: n^2 doubles must fit the 160 KiB LDS
constexpr int kCoarseChunks = 32;   // node chunks per aggregate in the Galerkin-product reduction
constexpr int kSlicePad = 16;       // per-(slice,kpos) column-index record length (>= rows per slice: 7 or 9)

// Device scalars of one BiCGStab solve (one cache line; written by designated lanes, read after a kernel boundary).
struct KrylovScalars {
  double rho;        // (rhat, r) of the iteration that just finished its update
  double rho_next;   // (rhat, r) computed by vec1 for the iteration in flight
  double alpha;
  double tol;        // absolute threshold on ||r||_2
  double rr;         // ||r||_2^2 seen by the last convergence test
  int32_t iters;     // completed iterations
  int32_t max_iters;
  int32_t done;      // 1 = converged, 2 = max_iters, 3 = breakdown
  int32_t it_cur;    // copy of iters made by vec1 (vec2 reads this one: it rewrites iters itself)
};

// Everything the kernels need, passed by value (kernarg segment).
struct Ctx {
  // sizes
  int32_t nv, nc, ndof, nb, nslices, n_work, nagg, ncoarse, n_vecwg, ncp;
  int32_t n_robin;
  // model / quadrature (device copies)
  const gmpnp_model_t* model;
  const gmpnp_quadrature_t* quad;
  // mesh (internal vertex order)
  const double* coords;   // [nv][DIM]
  const int32_t* cells;   // [nc][NN]
  // state
  double* u;
  double* un;
  double* F;
  // Dirichlet
  const uint8_t* bcflag;  // [ndof]
  const double* bcval;    // [ndof]
  // boundary terms
  const double* bndF;        // [ndof] constant part of the facet/point integrals
  const int32_t* robF_ptr;   // [ndof+1] CSR over rows of the Robin mass entries
  const int32_t* rob_col;    // [n_robin] column dof
  const double* rob_val;     // [n_robin]
  const int64_t* rob_addr;   // [n_robin] address in vals
  const int32_t* rob_row;    // [n_robin] row dof
  // element intermediates
  double* EF;   // [nc][NN*NF]
  double* EJ;   // [nc][EJ_STRIDE]
  // node -> (element, local node) incidence
  const int32_t* n2e_ptr;  // [nv+1]
  const int32_t* n2e;      // e*NN + a
  // BSR pattern (internal order), contributions per block
  const int32_t* rowptr;   // [nv+1]
  const int32_t* cols;     // [nb]
  const int32_t* cptr;     // [nb+1]
  const int32_t* contrib;  // e*16 + a*4 + b
  // SELL-(rows per slice)-storage of the Jacobian: vals[slice_off[s] + (kpos*NF + j)*64 + lane], lane = Iloc*NF + i
  double* vals;
  const int64_t* slice_off;     // [nslices+1] in doubles
  const int32_t* slice_colbase; // [nslices+1]
  const int32_t* sell_cols;     // [(colbase+kpos)*kSlicePad + Iloc] column node | aggregate << 24 (padding: the row's own node)
  const int32_t* sell_blk;      // same indexing: BSR block index, -1 = padding
  const uint8_t* sell_aggslot;  // same indexing: slot of the column's aggregate in row_aggs[I] (255 = padding)
  const int32_t* wl_slice;      // [n_work]
  const int32_t* wl_kpos;       // [n_work]
  // preconditioner
  double* Dinv;                 // [nv][NF][NF]
  const int32_t* agg;           // [nv]
  const int32_t* agg_start;     // [nagg+1] node ranges
  const int32_t* row_aggs;      // [nv][kMaxRowAggs]
  double* AP;                   // [ndof][kMaxRowAggs][NF]
  double* AcPart;               // [kCoarseChunks][ncoarse][ncoarse] partial sums of Ac
  double* Ac;                   // [ncoarse][ncoarse]
  double* AciT;                 // transposed inverse
  // vector-kernel workgroup table (aggregate- and node-aligned)
  const int32_t* vw_node0;      // [n_vecwg]
  const int32_t* vw_node1;
  const int32_t* agg_vw_ptr;    // [nagg+1]
  const int32_t* vw_agg;        // [n_vecwg] aggregate of each vector workgroup
  int32_t vw_slots;             // partial-sum slots per coarse dof (>= workgroups per aggregate)
  // Krylov vectors
  double* kr;    // r
  double* krhat;
  double* kp;
  double* kv;
  double* ks;
  double* kt;
  double* ky;
  double* kq;    // Dinv * (p or s)
  double* pc_part;  // [ncoarse][vw_slots] restriction partials, slot = workgroup index inside its aggregate
  double* yc;       // [ncoarse]
  double* part_rr;  // [n_vecwg]
  double* part_a;   // [nslices]      (rhat, v)
  double* part_b;   // [4][nslices]   (t,s) (t,t) (rhat,s) (rhat,t)
  double* part_f;   // residual-norm partials
  KrylovScalars* scal;
  int32_t* status;  // device error flags (bit 0: 1-S<=0, bit 1: singular block, bit 2: singular coarse)
};

// Host-side topology/layout tables (internal vertex order).
struct Topology {
  int dim = 0, nf = 0, nn = 0, nv = 0, nc = 0, S = 0;
  std::vector<int32_t> perm, iperm;       // perm[internal] = file ; iperm[file] = internal
  std::vector<double> coords;
  std::vector<int32_t> cells;
  std::vector<int32_t> rowptr, cols, cptr, contrib, n2e_ptr, n2e;
  std::vector<int32_t> sellk;      // [nb] position of BSR block k inside its SELL row (diagonal block: 0)
  int nslices = 0;
  std::vector<int64_t> slice_off;
  std::vector<int32_t> slice_colbase, sell_cols, sell_blk, wl_slice, wl_kpos;
  std::vector<uint8_t> sell_aggslot;
  int nagg = 0;
  std::vector<int32_t> agg, agg_start, row_aggs;
  std::vector<int32_t> vw_node0, vw_node1, agg_vw_ptr, vw_agg;
  int vw_slots = 0;
};

// Builds every table above; returns an error message or "" on success.
std::string build_topology(const gmpnp_mesh_t& mesh, int nf, int n_aggregates_requested, Topology& t);

}  // namespace gmpnp
