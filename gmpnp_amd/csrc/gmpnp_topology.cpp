// Host-side construction of the device layout tables: internal vertex order, node-block BSR pattern,
// per-block element contribution lists (race-free gather assembly), SELL slices, slab aggregates.
// Plays the role of DOLFIN's dofmap / sparsity-pattern builder ([3P]).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "gmpnp_internal.h"

namespace gmpnp {

static std::string build_pass(const gmpnp_mesh_t& m, const int32_t* perm_in, int nf, int nagg_req, bool fixed_nagg, Topology& t,
                              const gmpnp_partition_t* part);

// Two passes: the first (caller's order) fixes the aggregate count and the node degrees; nodes are then stably
// re-sorted by decreasing degree INSIDE each aggregate (equal-length rows per SELL slice, aggregates stay the exact
// slabs of the caller's order) and everything is rebuilt in that order.
std::string build_topology(const gmpnp_mesh_t& m, int nf, int nagg_req, Topology& t, const gmpnp_partition_t* part) {
  Topology first;
  std::string err = build_pass(m, m.perm, nf, nagg_req, false, first, part);
  if (!err.empty()) return err;
  std::vector<int32_t> perm2(first.perm);
  if (m.dim != 1)  // interval meshes keep the caller's path order (block-tridiagonal direct solver)
  for (int g = 0; g < first.nagg; ++g)
    std::stable_sort(perm2.begin() + first.agg_start[g], perm2.begin() + first.agg_start[g + 1], [&](int a, int b) {
      const int ia = first.iperm[a], ib = first.iperm[b];
      return (first.rowptr[ia + 1] - first.rowptr[ia]) > (first.rowptr[ib + 1] - first.rowptr[ib]);
    });
  err = build_pass(m, perm2.data(), nf, first.nagg, true, t, part);
  if (!err.empty()) return err;
  // Elimination order of the block-banded direct solver: the caller's slab order (sorted along the pore axis), NOT the
  // degree-sorted internal order, whose bandwidth is a whole aggregate.
  t.lu_node.resize(t.nv); t.lu_pos.resize(t.nv);
  for (int p = 0; p < t.nv; ++p) { const int I = t.iperm[first.perm[p]]; t.lu_node[p] = I; t.lu_pos[I] = p; }
  t.lu_band = 0;
  for (int I = 0; I < t.nv; ++I)
    for (int k = t.rowptr[I]; k < t.rowptr[I + 1]; ++k) t.lu_band = std::max(t.lu_band, std::abs(t.lu_pos[I] - t.lu_pos[t.cols[k]]));
  return "";
}

static std::string build_pass(const gmpnp_mesh_t& m, const int32_t* perm_in, int nf, int nagg_req, bool fixed_nagg, Topology& t,
                              const gmpnp_partition_t* part) {
  if (m.dim != 1 && m.dim != 3) return "mesh.dim must be 1 or 3";
  if (m.n_vertices <= 0 || m.n_cells <= 0 || !m.coords || !m.cells) return "empty mesh";
  t.dim = m.dim; t.nf = nf; t.nn = m.dim + 1; t.nv = m.n_vertices; t.nc = m.n_cells;
  t.S = kWave / nf;
  const int nv = t.nv, nc = t.nc, nn = t.nn, dim = t.dim;

  // ---- internal order -----------------------------------------------------------------------
  t.perm.resize(nv); t.iperm.assign(nv, -1);
  for (int i = 0; i < nv; ++i) {
    int f = perm_in ? perm_in[i] : i;
    if (f < 0 || f >= nv || t.iperm[f] != -1) return "perm is not a permutation";
    t.perm[i] = f; t.iperm[f] = i;
  }
  t.coords.resize((size_t)nv * dim);
  for (int i = 0; i < nv; ++i)
    for (int d = 0; d < dim; ++d) t.coords[(size_t)i * dim + d] = m.coords[(size_t)t.perm[i] * dim + d];
  t.cells.resize((size_t)nc * nn);
  for (size_t k = 0; k < (size_t)nc * nn; ++k) {
    int v = m.cells[k];
    if (v < 0 || v >= nv) return "cell references a vertex outside the mesh";
    t.cells[k] = t.iperm[v];
  }
  for (int e = 0; e < nc; ++e)
    for (int a = 0; a < nn; ++a)
      for (int b = a + 1; b < nn; ++b)
        if (t.cells[(size_t)e * nn + a] == t.cells[(size_t)e * nn + b]) return "degenerate cell (repeated vertex)";

  // ---- node adjacency -> BSR pattern ----------------------------------------------------------
  std::vector<int64_t> pairs; pairs.reserve((size_t)nc * nn * nn + nv);
  for (int e = 0; e < nc; ++e)
    for (int a = 0; a < nn; ++a)
      for (int b = 0; b < nn; ++b)
        pairs.push_back((int64_t)t.cells[(size_t)e * nn + a] * nv + t.cells[(size_t)e * nn + b]);
  for (int i = 0; i < nv; ++i) pairs.push_back((int64_t)i * nv + i);  // isolated vertices keep a diagonal
  std::sort(pairs.begin(), pairs.end());
  pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
  const int nb = (int)pairs.size();
  t.rowptr.assign(nv + 1, 0); t.cols.resize(nb);
  for (int k = 0; k < nb; ++k) { t.rowptr[pairs[k] / nv + 1]++; t.cols[k] = (int32_t)(pairs[k] % nv); }
  for (int i = 0; i < nv; ++i) t.rowptr[i + 1] += t.rowptr[i];
  auto find_block = [&](int I, int J) {
    const int32_t* b = t.cols.data() + t.rowptr[I]; const int32_t* e = t.cols.data() + t.rowptr[I + 1];
    return (int)(std::lower_bound(b, e, J) - t.cols.data());
  };

  // ---- contributions per block (element order => deterministic summation) ---------------------
  t.cptr.assign(nb + 1, 0);
  std::vector<int32_t> blk_of((size_t)nc * nn * nn);
  for (int e = 0; e < nc; ++e)
    for (int a = 0; a < nn; ++a)
      for (int b = 0; b < nn; ++b) {
        int k = find_block(t.cells[(size_t)e * nn + a], t.cells[(size_t)e * nn + b]);
        blk_of[((size_t)e * nn + a) * nn + b] = k; t.cptr[k + 1]++;
      }
  for (int k = 0; k < nb; ++k) t.cptr[k + 1] += t.cptr[k];
  t.contrib.resize(t.cptr[nb]);
  { std::vector<int32_t> fill(t.cptr.begin(), t.cptr.end() - 1);
    for (int e = 0; e < nc; ++e)
      for (int a = 0; a < nn; ++a)
        for (int b = 0; b < nn; ++b)
          t.contrib[fill[blk_of[((size_t)e * nn + a) * nn + b]]++] = e * 16 + a * 4 + b; }

  // ---- node -> (element, local node) ----------------------------------------------------------
  t.n2e_ptr.assign(nv + 1, 0);
  for (size_t k = 0; k < (size_t)nc * nn; ++k) t.n2e_ptr[t.cells[k] + 1]++;
  for (int i = 0; i < nv; ++i) t.n2e_ptr[i + 1] += t.n2e_ptr[i];
  t.n2e.resize(t.n2e_ptr[nv]);
  { std::vector<int32_t> fill(t.n2e_ptr.begin(), t.n2e_ptr.end() - 1);
    for (int e = 0; e < nc; ++e)
      for (int a = 0; a < nn; ++a) t.n2e[fill[t.cells[(size_t)e * nn + a]]++] = e * nn + a; }

  // ---- aggregates: contiguous, equal-count ranges of the internal order -------------------------
  int nagg_max = std::min(kMaxCoarse / nf, 16);  // LDS-resident block Gauss-Jordan must fit in 160 KiB; TileCoarse sums <= 16 column blocks
  while (nagg_max > 1 && (size_t)((nagg_max * nf) * (nagg_max * nf) + nagg_max * nf * nf + 2 * nf * nf) * sizeof(double) > 160u * 1024u) --nagg_max;
  auto fill_row_aggs = [&]() {  // aggregates each block row touches (at most kMaxRowAggs)
    t.row_aggs.assign((size_t)nv * kMaxRowAggs, -1);
    for (int i = 0; i < nv; ++i) {
      int cnt = 0;
      for (int k = t.rowptr[i]; k < t.rowptr[i + 1]; ++k) {
        int g = t.agg[t.cols[k]]; bool seen = false;
        for (int q = 0; q < cnt; ++q) seen |= (t.row_aggs[(size_t)i * kMaxRowAggs + q] == g);
        if (!seen) { if (cnt == kMaxRowAggs) return false; t.row_aggs[(size_t)i * kMaxRowAggs + cnt++] = g; }
      }
    }
    return true;
  };
  int nagg;
  t.own_node0 = 0; t.own_node1 = nv; t.own_agg0 = 0;
  if (part) {
    // Mesh partition (one handle per rank): the caller numbers the coarse slabs over the WHOLE mesh and tells every
    // local vertex its slab; the internal order must run through the slabs in ascending order (a slab order of the
    // local vertices does).  Owned vertices form one contiguous range of whole slabs; ghost vertices lie in slabs
    // owned by the neighbours, so every tile of the Krylov kernels is either all owned or all ghost.
    nagg = part->n_global_aggregates;
    if (nagg < 1 || nagg > nagg_max) return "partition: n_global_aggregates out of range (coarse operator must fit the LDS-resident inverse)";
    if (!part->vertex_aggregate || !part->vertex_owned) return "partition: vertex_aggregate / vertex_owned missing";
    t.nagg = nagg; t.agg.resize(nv); t.agg_start.assign(nagg + 1, 0);
    int prev = 0, first_own = -1, last_own = -1;
    for (int i = 0; i < nv; ++i) {
      const int g = part->vertex_aggregate[t.perm[i]];
      if (g < 0 || g >= nagg) return "partition: vertex_aggregate out of range";
      if (g < prev) return "partition: the vertex order (perm) must run through the aggregates in ascending order";
      prev = g; t.agg[i] = g; t.agg_start[g + 1]++;
      if (part->vertex_owned[t.perm[i]]) { if (first_own < 0) first_own = i; last_own = i; }
    }
    for (int g = 0; g < nagg; ++g) t.agg_start[g + 1] += t.agg_start[g];
    if (first_own < 0) return "partition: no owned vertex";
    for (int i = first_own; i <= last_own; ++i) if (!part->vertex_owned[t.perm[i]]) return "partition: owned vertices must be contiguous in the vertex order";
    t.own_node0 = first_own; t.own_node1 = last_own + 1;
    const int ga = t.agg[first_own], gb = t.agg[last_own];
    if (t.agg_start[ga] != first_own || t.agg_start[gb + 1] != last_own + 1) return "partition: an aggregate mixes owned and ghost vertices";
    t.own_agg0 = ga; t.own_agg1 = gb + 1;
    if (!fill_row_aggs()) return "partition: a block row touches more than kMaxRowAggs aggregates";
  } else {
  // default: 8 slabs.  On the pore meshes 8..60 slabs give the same Krylov iteration count (the block-Jacobi smoother
  // limits convergence, tools/precond_experiment.py), fewer than 8 lose it, and the coarse set-up cost grows with nagg^3.
  nagg = nagg_req > 0 ? std::min(nagg_req, nagg_max) : std::min(nagg_max, 8);
  nagg = std::max(1, std::min(nagg, nv / 8 > 0 ? nv / 8 : 1));
  if (fixed_nagg) nagg = nagg_req;
  for (;; --nagg) {  // shrink until no row touches more than kMaxRowAggs aggregates
    t.nagg = nagg; t.agg.resize(nv); t.agg_start.assign(nagg + 1, 0);
    for (int g = 0; g <= nagg; ++g) t.agg_start[g] = (int32_t)((int64_t)nv * g / nagg);
    for (int g = 0; g < nagg; ++g)
      for (int i = t.agg_start[g]; i < t.agg_start[g + 1]; ++i) t.agg[i] = g;
    const bool ok = fill_row_aggs();
    if (ok || nagg == 1) break;
    if (fixed_nagg) return "internal error: aggregate count changed between passes";
  }
  nagg = t.nagg;
  t.own_agg1 = nagg;
  }

  // ---- SELL slices ------------------------------------------------------------------------------
  // Inside a row the blocks may sit in any order: put the diagonal first and the others by decreasing
  // number of element contributions, so the rows of one slice present the gather kernel with similar work.
  if (nv >= (1 << 24)) return "mesh too large for the packed (column, aggregate) index";
  t.sellk.resize(nb);
  { std::vector<int> ord;
    for (int I = 0; I < nv; ++I) {
      ord.clear();
      for (int k = t.rowptr[I]; k < t.rowptr[I + 1]; ++k) ord.push_back(k);
      std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) {
        const bool dx = t.cols[x] == I, dy = t.cols[y] == I;
        if (dx != dy) return dx;
        return (t.cptr[x + 1] - t.cptr[x]) > (t.cptr[y + 1] - t.cptr[y]);
      });
      for (size_t q = 0; q < ord.size(); ++q) t.sellk[ord[q]] = (int32_t)q;
    } }
  // Slices are aggregate-aligned: the nodes of aggregate g are cut into slices of S nodes (the last one may be
  // partial), consecutive kSlicesPerTile slices of one aggregate form a TILE = the row range of one Krylov workgroup,
  // so the coarse restriction of a tile's rows is one partial sum into one aggregate.
  const int S = t.S;
  t.slice_node0.clear(); t.slice_nn.clear(); t.node_slice.assign(nv, 0);
  t.tile_slice0.clear(); t.tile_agg.clear(); t.tile_slot.clear(); t.agg_tile_ptr.assign(nagg + 1, 0);
  for (int g = 0; g < nagg; ++g) {
    int slot = 0;
    for (int n0 = t.agg_start[g]; n0 < t.agg_start[g + 1]; n0 += S) {
      const int sidx = (int)t.slice_node0.size();
      if ((sidx - (t.tile_slice0.empty() ? 0 : t.tile_slice0.back())) >= kSlicesPerTile || t.tile_agg.empty() ||
          t.tile_agg.back() != g) {
        t.tile_slice0.push_back(sidx); t.tile_agg.push_back(g); t.tile_slot.push_back(slot++);
      }
      const int nn_ = std::min(S, t.agg_start[g + 1] - n0);
      t.slice_node0.push_back(n0); t.slice_nn.push_back(nn_);
      for (int q = 0; q < nn_; ++q) t.node_slice[n0 + q] = sidx;
    }
    t.agg_tile_ptr[g + 1] = (int32_t)t.tile_slice0.size();
    t.tile_slots = std::max(t.tile_slots, slot);
  }
  t.nslices = (int)t.slice_node0.size();
  t.ntiles = (int)t.tile_slice0.size();
  t.own_tile0 = t.agg_tile_ptr[t.own_agg0]; t.own_ntiles = t.agg_tile_ptr[t.own_agg1] - t.own_tile0;
  t.tile_slice0.push_back(t.nslices);
  t.tile_slots = ((t.tile_slots + 7) / 8) * 8;
  t.slice_off.assign(t.nslices + 1, 0); t.slice_colbase.assign(t.nslices + 1, 0);
  for (int s = 0; s < t.nslices; ++s) {
    int mx = 0;
    for (int I = t.slice_node0[s]; I < t.slice_node0[s] + t.slice_nn[s]; ++I) mx = std::max(mx, t.rowptr[I + 1] - t.rowptr[I]);
    t.slice_colbase[s + 1] = t.slice_colbase[s] + mx;
    t.slice_off[s + 1] = t.slice_off[s] + (int64_t)mx * nf * kWave;
  }
  // aggregates each tile prolongs from (its rows' column nodes): the fused Krylov kernels evaluate only these coarse rows
  t.tile_aggs.assign((size_t)t.ntiles * kTileAggs, 0); t.tile_nagg.assign(t.ntiles, 0);
  for (int tl = 0; tl < t.ntiles; ++tl)
    for (int s = t.tile_slice0[tl]; s < t.tile_slice0[tl + 1]; ++s)
      for (int I = t.slice_node0[s]; I < t.slice_node0[s] + t.slice_nn[s]; ++I)
        for (int q = 0; q < kMaxRowAggs; ++q) {
          const int g = t.row_aggs[(size_t)I * kMaxRowAggs + q];
          if (g < 0) continue;
          bool seen = false;
          for (int z = 0; z < t.tile_nagg[tl]; ++z) seen |= (t.tile_aggs[(size_t)tl * kTileAggs + z] == g);
          if (!seen) {
            if (t.tile_nagg[tl] == kTileAggs) return "a Krylov tile touches more than kTileAggs coarse aggregates";
            t.tile_aggs[(size_t)tl * kTileAggs + t.tile_nagg[tl]++] = g;
          }
        }
  // Gather work list: one wave per (slice, block position).  The list is cut into runs of at most 80 contiguous slices, a
  // multiple of 8 of them; XCD x (= workgroup index mod 8) works through runs x, x + 8, ... one after the other
  // (k_jac_gather / k_scale_columns: xcd_run_wave), so the element records around a run's nodes are shared through ONE
  // XCD's L2 while they are hot, and inside a run the heaviest positions (diagonal blocks: most element contributions) come
  // first.  Runs are padded to equal length with (-1, 0) entries (a wave that reads slice -1 exits).
  { const int kRuns = 8 * std::max(1, (t.nslices + 639) / 640);
    constexpr int kWavesPerBlock = kVecBlock / kWave;
    std::vector<std::vector<std::pair<int, int>>> runs(kRuns);
    size_t longest = 0;
    for (int g = 0; g < kRuns; ++g) {
      const int s0 = (int)((int64_t)t.nslices * g / kRuns), s1 = (int)((int64_t)t.nslices * (g + 1) / kRuns);
      int mxall = 0;
      for (int s = s0; s < s1; ++s) mxall = std::max(mxall, t.slice_colbase[s + 1] - t.slice_colbase[s]);
      for (int kp = 0; kp < mxall; ++kp)
        for (int s = s0; s < s1; ++s)
          if (kp < t.slice_colbase[s + 1] - t.slice_colbase[s]) runs[g].push_back({s, kp});
      longest = std::max(longest, runs[g].size());
    }
    longest = ((longest + kWavesPerBlock - 1) / kWavesPerBlock) * kWavesPerBlock;
    t.wl_run_blocks = (int)(longest / kWavesPerBlock);
    for (int g = 0; g < kRuns; ++g)
      for (size_t q = 0; q < longest; ++q) {
        const bool on = q < runs[g].size();
        t.wl_slice.push_back(on ? runs[g][q].first : -1); t.wl_kpos.push_back(on ? runs[g][q].second : 0);
      } }
  const int ncolrec = t.slice_colbase[t.nslices];
  t.sell_cols.assign((size_t)ncolrec * kSlicePad, 0); t.sell_aggslot.assign((size_t)ncolrec * kSlicePad, 255);
  t.sell_blk.assign((size_t)ncolrec * kSlicePad, -1);
  std::vector<int> tile_of_slice(t.nslices);
  for (int tl = 0; tl < t.ntiles; ++tl)
    for (int s = t.tile_slice0[tl]; s < t.tile_slice0[tl + 1]; ++s) tile_of_slice[s] = tl;
  // distinct column nodes per tile (the Krylov kernels stage x for exactly these in LDS) and the tile-local index of
  // every block's column
  t.tile_colptr.assign(t.ntiles + 1, 0); t.tile_cols.clear(); t.tile_colslot.clear();
  t.sell_lcol.assign((size_t)(ncolrec + kRowPad) * kSlicePad, 0);  // + padding for the unconditional preload
  { std::vector<int> where(nv, -1), touched;
    for (int tl = 0; tl < t.ntiles; ++tl) {
      touched.clear();
      for (int s = t.tile_slice0[tl]; s < t.tile_slice0[tl + 1]; ++s)
        for (int I = t.slice_node0[s]; I < t.slice_node0[s] + t.slice_nn[s]; ++I)
          for (int k = t.rowptr[I]; k < t.rowptr[I + 1]; ++k) {
            const int J = t.cols[k];
            if (where[J] < 0) { where[J] = 0; touched.push_back(J); }
          }
      std::sort(touched.begin(), touched.end());
      if (touched.size() > (size_t)kTileCols) return "a Krylov tile references more than kTileCols column nodes";
      for (size_t q = 0; q < touched.size(); ++q) {
        where[touched[q]] = (int)q;
        t.tile_cols.push_back(touched[q]);
        t.tile_colslot.push_back(t.agg[touched[q]]);
      }
      for (int s = t.tile_slice0[tl]; s < t.tile_slice0[tl + 1]; ++s)
        for (int I = t.slice_node0[s]; I < t.slice_node0[s] + t.slice_nn[s]; ++I) {
          const int il = I - t.slice_node0[s], mx = t.slice_colbase[s + 1] - t.slice_colbase[s];
          for (int kp = 0; kp < mx; ++kp) t.sell_lcol[((size_t)t.slice_colbase[s] + kp) * kSlicePad + il] = where[I];  // padding
          for (int k = t.rowptr[I]; k < t.rowptr[I + 1]; ++k)
            t.sell_lcol[((size_t)t.slice_colbase[s] + t.sellk[k]) * kSlicePad + il] = where[t.cols[k]];
        }
      for (int J : touched) where[J] = -1;
      t.tile_colptr[tl + 1] = (int32_t)t.tile_cols.size();
    } }
  { static_assert(kSlicesPerTile == 1, "tile records describe one slice");
    int mxc = 0;
    for (int tl = 0; tl < t.ntiles; ++tl) mxc = std::max(mxc, t.tile_colptr[tl + 1] - t.tile_colptr[tl]);
    t.col_stride = ((mxc + 15) / 16) * 16;
    std::vector<int32_t> fc((size_t)t.ntiles * t.col_stride, 0), fs((size_t)t.ntiles * t.col_stride, 0);
    t.tile_rec.resize(t.ntiles);
    for (int tl = 0; tl < t.ntiles; ++tl) {
      const int c0 = t.tile_colptr[tl], nc_ = t.tile_colptr[tl + 1] - c0, s0 = t.tile_slice0[tl];
      for (int q = 0; q < nc_; ++q) { fc[(size_t)tl * t.col_stride + q] = t.tile_cols[c0 + q]; fs[(size_t)tl * t.col_stride + q] = t.tile_colslot[c0 + q]; }
      TileRec r{};
      r.slice_off = t.slice_off[s0]; r.colbase = t.slice_colbase[s0]; r.mx = t.slice_colbase[s0 + 1] - t.slice_colbase[s0];
      r.node0 = t.slice_node0[s0]; r.nn = t.slice_nn[s0]; r.ncols = nc_; r.agg = t.tile_agg[tl]; r.slot = t.tile_slot[tl];
      t.tile_rec[tl] = r;
    }
    t.tile_cols.swap(fc); t.tile_colslot.swap(fs); }
  for (int I = 0; I < nv; ++I) {
    const int s = t.node_slice[I], il = I - t.slice_node0[s], tl = tile_of_slice[s];
    const int mx = t.slice_colbase[s + 1] - t.slice_colbase[s];
    auto slot_of = [&](int g) {
      for (int z = 0; z < t.tile_nagg[tl]; ++z) if (t.tile_aggs[(size_t)tl * kTileAggs + z] == g) return z;
      return 0;
    };
    for (int kp = 0; kp < mx; ++kp)  // padding: value stays zero, the index stays valid
      t.sell_cols[((size_t)t.slice_colbase[s] + kp) * kSlicePad + il] = I | (slot_of(t.agg[I]) << 24);
    for (int k = t.rowptr[I]; k < t.rowptr[I + 1]; ++k) {
      const size_t rec = ((size_t)t.slice_colbase[s] + t.sellk[k]) * kSlicePad + il;
      const int J = t.cols[k];
      t.sell_cols[rec] = J | (slot_of(t.agg[J]) << 24);
      t.sell_blk[rec] = k;
      for (int q = 0; q < kMaxRowAggs; ++q)
        if (t.row_aggs[(size_t)I * kMaxRowAggs + q] == t.agg[J]) t.sell_aggslot[rec] = (uint8_t)q;
    }
  }
  return "";
}

}  // namespace gmpnp
