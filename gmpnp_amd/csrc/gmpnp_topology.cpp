// Host-side construction of the device layout tables: internal vertex order, node-block BSR pattern,
// per-block element contribution lists (race-free gather assembly), SELL slices, slab aggregates.
// Plays the role of DOLFIN's dofmap / sparsity-pattern builder ([3P]).
#include <algorithm>
#include <cstring>
#include <numeric>

#include "gmpnp_internal.h"

namespace gmpnp {

std::string build_topology(const gmpnp_mesh_t& m, int nf, int nagg_req, Topology& t) {
  if (m.dim != 1 && m.dim != 3) return "mesh.dim must be 1 or 3";
  if (m.n_vertices <= 0 || m.n_cells <= 0 || !m.coords || !m.cells) return "empty mesh";
  t.dim = m.dim; t.nf = nf; t.nn = m.dim + 1; t.nv = m.n_vertices; t.nc = m.n_cells;
  t.S = kWave / nf;
  const int nv = t.nv, nc = t.nc, nn = t.nn, dim = t.dim;

  // ---- internal order -----------------------------------------------------------------------
  t.perm.resize(nv); t.iperm.assign(nv, -1);
  for (int i = 0; i < nv; ++i) {
    int f = m.perm ? m.perm[i] : i;
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
  int nagg_max = kMaxCoarse / nf;  // and the LDS-resident block Gauss-Jordan must fit in 160 KiB
  while (nagg_max > 1 && (size_t)((nagg_max * nf) * (nagg_max * nf) + nagg_max * nf * nf + 2 * nf * nf) * sizeof(double) > 160u * 1024u) --nagg_max;
  int nagg = nagg_req > 0 ? std::min(nagg_req, nagg_max) : nagg_max;
  nagg = std::max(1, std::min(nagg, nv / 8 > 0 ? nv / 8 : 1));
  for (;; --nagg) {  // shrink until no row touches more than kMaxRowAggs aggregates
    t.nagg = nagg; t.agg.resize(nv); t.agg_start.assign(nagg + 1, 0);
    for (int g = 0; g <= nagg; ++g) t.agg_start[g] = (int32_t)((int64_t)nv * g / nagg);
    for (int g = 0; g < nagg; ++g)
      for (int i = t.agg_start[g]; i < t.agg_start[g + 1]; ++i) t.agg[i] = g;
    t.row_aggs.assign((size_t)nv * kMaxRowAggs, -1);
    bool ok = true;
    for (int i = 0; i < nv && ok; ++i) {
      int cnt = 0;
      for (int k = t.rowptr[i]; k < t.rowptr[i + 1]; ++k) {
        int g = t.agg[t.cols[k]]; bool seen = false;
        for (int q = 0; q < cnt; ++q) seen |= (t.row_aggs[(size_t)i * kMaxRowAggs + q] == g);
        if (!seen) { if (cnt == kMaxRowAggs) { ok = false; break; } t.row_aggs[(size_t)i * kMaxRowAggs + cnt++] = g; }
      }
    }
    if (ok || nagg == 1) break;
  }
  nagg = t.nagg;

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
  const int S = t.S;
  t.nslices = (nv + S - 1) / S;
  t.slice_off.assign(t.nslices + 1, 0); t.slice_colbase.assign(t.nslices + 1, 0);
  for (int s = 0; s < t.nslices; ++s) {
    int mx = 0;
    for (int I = s * S; I < std::min(nv, (s + 1) * S); ++I) mx = std::max(mx, t.rowptr[I + 1] - t.rowptr[I]);
    t.slice_colbase[s + 1] = t.slice_colbase[s] + mx;
    t.slice_off[s + 1] = t.slice_off[s] + (int64_t)mx * nf * kWave;
  }
  // gather work list, heaviest SELL positions (diagonal blocks) first
  { int mxall = 0;
    for (int s = 0; s < t.nslices; ++s) mxall = std::max(mxall, t.slice_colbase[s + 1] - t.slice_colbase[s]);
    for (int kp = 0; kp < mxall; ++kp)
      for (int s = 0; s < t.nslices; ++s)
        if (kp < t.slice_colbase[s + 1] - t.slice_colbase[s]) { t.wl_slice.push_back(s); t.wl_kpos.push_back(kp); } }
  const int ncolrec = t.slice_colbase[t.nslices];
  t.sell_cols.assign((size_t)ncolrec * kSlicePad, 0); t.sell_aggslot.assign((size_t)ncolrec * kSlicePad, 255);
  t.sell_blk.assign((size_t)ncolrec * kSlicePad, -1);
  for (int I = 0; I < nv; ++I) {
    const int s = I / S, il = I - s * S;
    const int mx = t.slice_colbase[s + 1] - t.slice_colbase[s];
    for (int kp = 0; kp < mx; ++kp)  // padding: value stays zero, the index stays valid
      t.sell_cols[((size_t)t.slice_colbase[s] + kp) * kSlicePad + il] = I | (t.agg[I] << 24);
    for (int k = t.rowptr[I]; k < t.rowptr[I + 1]; ++k) {
      const size_t rec = ((size_t)t.slice_colbase[s] + t.sellk[k]) * kSlicePad + il;
      const int J = t.cols[k];
      t.sell_cols[rec] = J | (t.agg[J] << 24);
      t.sell_blk[rec] = k;
      for (int q = 0; q < kMaxRowAggs; ++q)
        if (t.row_aggs[(size_t)I * kMaxRowAggs + q] == t.agg[J]) t.sell_aggslot[rec] = (uint8_t)q;
    }
  }

  // ---- vector-kernel workgroups: whole nodes, one aggregate each --------------------------------
  const int npw = kVecBlock / nf;
  t.agg_vw_ptr.assign(nagg + 1, 0);
  for (int g = 0; g < nagg; ++g) {
    for (int n0 = t.agg_start[g]; n0 < t.agg_start[g + 1]; n0 += npw) {
      t.vw_node0.push_back(n0); t.vw_node1.push_back(std::min(n0 + npw, t.agg_start[g + 1])); t.vw_agg.push_back(g);
    }
    t.agg_vw_ptr[g + 1] = (int32_t)t.vw_node0.size();
    t.vw_slots = std::max(t.vw_slots, t.agg_vw_ptr[g + 1] - t.agg_vw_ptr[g]);
  }
  t.vw_slots = ((t.vw_slots + 15) / 16) * 16;
  return "";
}

}  // namespace gmpnp
