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
constexpr int kCoarseChunks = 32;   // node chunks per aggregate in the Galerkin-product reduction: the minimum (Ctx::coarse_chunks grows with the mesh)
#ifndef GMPNP_SLICES_PER_TILE
#define GMPNP_SLICES_PER_TILE 1
#endif
#ifndef GMPNP_KRYLOV_WAVES
#define GMPNP_KRYLOV_WAVES 8
#endif
#ifndef GMPNP_ROW_PRELOAD
#define GMPNP_ROW_PRELOAD 1
#endif
constexpr int kSlicesPerTile = GMPNP_SLICES_PER_TILE;    // SELL slices (of 7 or 9 block rows) per Krylov workgroup
constexpr int kTileAggs = 6;         // coarse aggregates the rows of one tile may prolong from
constexpr int kTileCols = 224;       // distinct column nodes a tile may reference (x staged in LDS: kTileCols*NF doubles)
constexpr int kSlicePad = 16;       // per-(slice,kpos) column-index record length (>= rows per slice: 7 or 9)
#ifndef GMPNP_ROW_PRELOAD_B
#define GMPNP_ROW_PRELOAD_B 1  // B half of the two-launch form: a second block position up front fits its registers and was the default until the hand-over got short — now it arrives late like in A (same box: B 9.68 -> 9.46 us, 664/674 -> 671/678 its/s with one position)
#endif
#ifndef GMPNP_CLAMP_PRELOAD
#define GMPNP_CLAMP_PRELOAD 1  // preload positions past a slice's end repeat its last position instead of reading the next slice
#endif
#ifndef GMPNP_ROW_PRELOAD_A
#define GMPNP_ROW_PRELOAD_A 1  // A half of the two-launch form (2 fits without spilling since the staged s, t wait in LDS, but measured slower: A 13.7 -> 14.1 us and B 11.7 -> 12.7 us)
#endif
constexpr int kRowPreMax = GMPNP_ROW_PRELOAD_B > GMPNP_ROW_PRELOAD_A ? (GMPNP_ROW_PRELOAD_B > GMPNP_ROW_PRELOAD ? GMPNP_ROW_PRELOAD_B : GMPNP_ROW_PRELOAD)
                                                                     : (GMPNP_ROW_PRELOAD_A > GMPNP_ROW_PRELOAD ? GMPNP_ROW_PRELOAD_A : GMPNP_ROW_PRELOAD);
constexpr int kRowPad = GMPNP_KRYLOV_WAVES * kRowPreMax;  // block positions of zero padding behind the last slice

// Device scalars of one BiCGStab solve.  Each field has ONE writer kernel and is read only by later launches
// (fields written by kernel A are read by B and vice versa; rho is double-buffered by iteration parity).
struct KrylovScalars {
  double rho[2];     // (rhat, r_k), slot k & 1, written by A(k)
  double alpha;      // written by coarse_b(k)
  double omega, beta; // written by coarse_a(k)
  double tol;        // absolute threshold on ||r||_2 (host)
  double rr;         // ||r||_2^2 seen by the last convergence test (B)
  double rr0;        // ||r_0||_2^2 of this pass (host): divergence guard
  int32_t iters;     // completed iterations (B)
  int32_t it_cur;    // unused (the iteration index is a kernel argument); kept for the struct layout
  int32_t max_iters; // host
  int32_t done;      // 1 = converged, 2 = max_iters, 3 = breakdown (published by B)
  int32_t done_next; // verdict of coarse_b(k), turned into `done` by B(k)
  int32_t pad_;
};

// Progress of the running BiCGStab solve in PINNED HOST memory (fine-grained): the B kernels write it with system-scope
// stores and the host reads it in a spin loop, instead of a device-to-host copy plus an event per polling burst.
struct HostPoll {
  double rr;      // ||r||^2 at the stopping test (valid once done != 0)
  int32_t iters;  // completed iterations (every B launch)
  int32_t done;   // exit code, written last
};

// Everything a Krylov workgroup needs to know about its tile (= one SELL slice), fetched with ONE load.
struct TileRec {
  int64_t slice_off;  // first value of the slice in vals / vals_s
  int32_t colbase;    // first (slice, kpos) record of sell_lcol
  int32_t mx;         // block positions in the slice
  int32_t node0, nn;  // rows
  int32_t ncols;      // distinct column nodes (listed at tile_cols + tile * col_stride)
  int32_t agg, slot;  // aggregate of the rows, partial-sum slot inside it
  int32_t pad_;
};

constexpr int kXFlagStride = 32;   // uint32 words between two ranks' flags in a peer mailbox (one 128-B line each)
constexpr int kLLSlots = 4;        // flagged-word areas of a peer mailbox: slots by sequence number (a reader looks one exchange back, a writer may be one ahead)
constexpr int kLLRow = 27;         // doubles per ghost node and slot there (three vectors of NF = 9)

// Everything the kernels need, passed by value (kernarg segment).
struct Ctx {
  // sizes
  int32_t nv, nc, ndof, nb, nslices, ntiles, n_work, nagg, ncoarse, tile_slots, n_robin, use_coarse;
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
  // SELL storage of the Jacobian: vals[slice_off[s] + (kpos*NF + j)*64 + lane], lane = Iloc*NF + i, node = slice_node0[s] + Iloc
  double* vals;     // A (assembled)
  double* vals_s;   // As = A Dinv (k_scale_columns), same layout
  const int64_t* slice_off;     // [nslices+1] in doubles
  const int32_t* slice_colbase; // [nslices+1]
  const int32_t* slice_node0;   // [nslices] first node of the slice (slices are aggregate-aligned)
  const int32_t* slice_nn;      // [nslices] nodes in the slice (<= S)
  const int32_t* node_slice;    // [nv]
  const int32_t* sell_cols;     // [(colbase+kpos)*kSlicePad + Iloc] column node | tile-local aggregate slot << 24 (padding: own node)
  const int32_t* sell_blk;      // same indexing: BSR block index, -1 = padding
  const uint8_t* sell_aggslot;  // same indexing: slot of the column's aggregate in row_aggs[I] (255 = padding)
  const int32_t* wl_slice;      // [n_work]
  const int32_t* wl_kpos;       // [n_work]
  // tiles = row ranges of the Krylov workgroups (kSlicesPerTile slices of ONE aggregate)
  const int32_t* tile_slice0;   // [ntiles+1]
  const int32_t* tile_agg;      // [ntiles]
  const int32_t* tile_slot;     // [ntiles] index of the tile inside its aggregate (partial-sum slot)
  const int32_t* tile_aggs;     // [ntiles][kTileAggs] aggregates the tile prolongs from
  const int32_t* tile_nagg;     // [ntiles]
  const TileRec* tile_rec;      // [ntiles]
  int32_t col_stride;           // fixed stride of the per-tile column lists below
  const int32_t* tile_cols;     // [ntiles][col_stride] distinct column nodes of the tile's rows (ascending; padding: node 0)
  const int32_t* tile_colslot;  // [ntiles][col_stride] aggregate of each of them
  const int32_t* sell_lcol;     // [(colbase+kpos)*kSlicePad + Iloc] index of the block's column in the tile's list
  // preconditioner
  double* Dinv;                 // [nv][NF][NF]
  const int32_t* agg;           // [nv]
  const int32_t* agg_start;     // [nagg+1] node ranges
  const int32_t* row_aggs;      // [nv][kMaxRowAggs]
  double* AP;                   // [ndof][kMaxRowAggs][NF]
  double* AcPart;               // [coarse_chunks][ncoarse][ncoarse] partial sums of Ac
  int32_t coarse_chunks;        // node chunks per aggregate in k_coarse_sum: 32 ... 1024, about 24 nodes each
  double* Ac;                   // [ncoarse][ncoarse]
  double* Aci;                  // inverse, row major
  // Krylov vectors (right-scaled system A Dinv (I + P Aci P^T) y = b)
  double* kr;
  double* krhat;
  double* kp[2];   // ping-pong: p_k lives in kp[k & 1]
  double* kv[2];
  double* ks;
  double* kt;
  double* ky;
  double* kx;      // work vector (plain SpMV input / M^{-1} output)
  double* kb;      // right-hand side of the linear solve in flight (k_res_gather leaves b = F in kr and kb)
  double* yc;      // [ncoarse] coarse solve of the half-iteration in flight
  // coarse level: per-tile partial restrictions written by the kernels' epilogues (double-buffered by iteration parity)
  double* cpart_r[2];  // P^T r_k (kernel A)
  double* cpart_p[2];  // P^T p_k (kernel A)
  double* cpart_v[2];  // [ncoarse][tile_slots] partial restriction of v (kernel A, by iteration parity) / k_restrict output
  double* cpart_t;  // [ncoarse][tile_slots] partial restriction of t (kernel B)
  double* part_rr;  // [ntiles]      ||r||^2 partials (A)
  double* part_a;   // [ntiles]      (rhat, v) (A)
  double* part_b;   // [4][ntiles]   (t,s) (t,t) (rhat,s) (rhat,t) (B)
  double* part_f;   // residual-norm partials: PINNED HOST memory (the host sums them after the stream sync, no copy)
  KrylovScalars* scal;
  int32_t* status;  // device error flags (bit 0: 1-S<=0, bit 1: singular block, bit 2: singular coarse, bit 3: hand-over timeout)
  uint32_t* ticket; // fused launch form: coarse workgroups finished so far in this solve
  HostPoll* poll;   // host-visible progress of the solve (pinned memory, device pointer)
  int32_t* status_host;  // pinned copy of *status, refreshed by k_res_gather (the host reads it with the residual norm)
  // partitioned solve (one handle per rank): owned ranges and the buffers of the fused exchanges
  int32_t own_node0, own_node1;   // owned nodes (internal order); [0, nv) for an unpartitioned handle
  int32_t own_agg0, own_agg1;     // aggregates made of owned nodes
  int32_t tile0;                  // first owned tile: the Krylov workgroups run tiles tile0 + blockIdx.x
  int32_t dist;                   // 1 = the coarse kernels read pre-reduced sums (red_i/a/b) instead of tile partials: partitioned
                                  //     solve (all-reduced over the ranks) and large unpartitioned meshes (k_dist_reduce per launch)
  int32_t wl_run_blocks;          // workgroups per run of the gather work list
  double* red_i;                  // [ncoarse]          P^T r_0                                  (all-reduced over the ranks)
  double* red_a;                  // [2 + 3 ncoarse]    (rhat,v) ||r||^2 | P^T v | P^T r | P^T p
  double* red_b;                  // [4 + ncoarse]      (t,s) (t,t) (rhat,s) (rhat,t) | P^T t
  // exchange as the PROLOGUE of a half-iteration's launch (peer transport, gmpnp_dist_kernels.h): the ranks' sums and boundary rows
  // arrive in THIS rank's mailbox as flagged words (every 8-byte word = 32 bits of data + the exchange's sequence number), in the
  // slot xseq & 3; the coarse workgroups and the boundary tiles poll the words they need — those of this launch's exchange (xseq)
  // and of the one before (xseq - 1).  nullptr otherwise.
  const unsigned long long* xll_red;    // [kLLSlots][xsize ranks][xcap][2 words]   contributions to the all-reduced sums
  const unsigned long long* xll_halo;   // [receive-list index][kLLSlots][kLLRow][2 words]   ghost rows (r|s, v|t, p)
  const int32_t* tile_cols_x;           // tile_cols with every ghost node replaced by -(k + 1), k = its index in the receive list
  uint32_t xseq;
  int32_t xsize, xcap;
  const double* stage_a;   // materialised form with a multilevel term: what half A / half B stage instead of p_k / s_k (else nullptr)
  const double* stage_b;
  const double* supg_rho;  // [nv][NS] nodal SUPG parameters (internal order) or nullptr: PNP stabilisation of reference 1D:597-722
  int32_t supg_w[GMPNP_MAX_SPECIES];  // species whose gradient enters species i's strong residual (identity except Q7)
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
  std::vector<int32_t> slice_node0, slice_nn, node_slice;            // aggregate-aligned slices
  std::vector<int32_t> tile_slice0, tile_agg, tile_slot, agg_tile_ptr, tile_aggs, tile_nagg;
  std::vector<int32_t> tile_colptr, tile_cols, tile_colslot, sell_lcol;  // tile_cols/tile_colslot: fixed stride col_stride
  std::vector<TileRec> tile_rec;
  int col_stride = 0;
  int ntiles = 0, tile_slots = 0;
  int wl_run_blocks = 0;                  // workgroups per run of the gather work list (wl_slice / wl_kpos)
  int own_node0 = 0, own_node1 = 0;       // internal node range this handle owns (partitioned solve; everything otherwise)
  int own_agg0 = 0, own_agg1 = 0;         // aggregates made of owned nodes
  int own_tile0 = 0, own_ntiles = 0;      // their tiles (contiguous)
  std::vector<int32_t> lu_node, lu_pos;   // elimination order of the block-banded LU: lu_node[position] = internal node
  int lu_band = 0;                        // max |lu_pos[I] - lu_pos[J]| over the blocks of the pattern
};

// Block-banded LU of the Jacobian in the elimination order above (direct fallback of the 3D Krylov solve).
// Block (p, q), |p - q| <= b, lives at band[(p * (2b+1) + (q - p + b)) * NF*NF], row major.
struct BandLU {
  double* band;           // [n][2b+1][NF][NF]
  double* dinv;           // [n][NF][NF] inverses of the pivot blocks
  const int32_t* lu_pos;  // [nv] internal node -> position
  const int32_t* lu_node; // [nv] position -> internal node
  int32_t n, b;
};

// Builds every table above; returns an error message or "" on success.
std::string build_topology(const gmpnp_mesh_t& mesh, int nf, int n_aggregates_requested, Topology& t,
                           const gmpnp_partition_t* part = nullptr);

}  // namespace gmpnp
