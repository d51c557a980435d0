// Mesh-partitioned Newton solve inside the library (include/gmpnp.h, "mesh-partitioned solve").
// Included at the end of gmpnp_api.hip: uses the handle type and the launch helpers defined there.
//
// One handle per rank on the rank's local mesh (owned + one ghost layer).  Per BiCGStab half-iteration and rank:
//     coarse kernel (scalars + coarse solve from all-reduced sums)  ->  tile kernel on the owned tiles (SpMV + vector updates)
//     ->  k_dist_reduce (per-rank sums)  ->  ONE all-reduce  +  ONE grouped send/recv of the ghost rows  ->  k_halo_unpack
// Nothing of the loop runs on the host except the launches; the host reads the device's verdict once per burst, and every
// rank launches the same bursts (the burst schedule depends only on all-reduced quantities), so the collectives pair up.
//
// Transports: peer mailboxes (one process per rank; every collective is ONE k_peer_exchange launch that stores into the other
// ranks' IPC-mapped mailboxes — over xGMI between GPUs — and waits on its own flags: gmpnp_dist_kernels.h); RCCL (ncclAllReduce /
// grouped ncclSend+ncclRecv on the solver's stream, librccl.so loaded with dlopen on first use); host-staged callbacks; or, for
// a group that holds ALL ranks of the partition in one process, device copies between the handles on one shared stream.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl_api(std::string* why) {
  static RcclApi api;
  static bool tried = false;
  static std::string err;
  if (!tried) {
    tried = true;
    // One RCCL per HIP runtime — and it has to be the one that sits on the HIP runtime THIS library is bound to.  A process
    // that imports PyTorch holds PyTorch's own libamdhip64.so and librccl.so next to the system's (same sonames): whichever
    // HIP runtime was loaded first serves this library, and an RCCL bound to the other one fails in ncclCommInitRank
    // ("unhandled cuda error").  So: find the file our HIP symbols come from and take the librccl.so in ITS directory
    // (torch/lib for PyTorch's pair, /opt/rocm/lib for the system's); by path, because a name or soname would match
    // whichever copy happens to be loaded already.
    std::string dir;
    { Dl_info info{};
      if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
        dir = info.dli_fname;
        const size_t slash = dir.find_last_of('/');
        dir = slash == std::string::npos ? std::string() : dir.substr(0, slash);
      } }
    if (!dir.empty())
      for (const char* name : {"/librccl.so", "/librccl.so.1"}) {
        api.lib = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
      }
    if (!api.lib)
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
      }
    if (!api.lib) err = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "?");
    else {
      auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); if (!p && err.empty()) err = std::string("librccl.so lacks ") + n; return p; };
      api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
      api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
      api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
      api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
      api.Send = (decltype(api.Send))sym("ncclSend");
      api.Recv = (decltype(api.Recv))sym("ncclRecv");
      api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
      api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
      api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    }
  }
  if (!err.empty()) { if (why) *why = err; return nullptr; }
  return &api;
}

#define NCCL_TRY(api, expr)                                                                        \
  do {                                                                                             \
    ncclResult_t r__ = (expr);                                                                     \
    if (r__ != ncclSuccess)                                                                        \
      return fail(GMPNP_ERR_HIP, std::string(#expr) + ": " + ((api)->GetErrorString ? (api)->GetErrorString(r__) : "RCCL error")); \
  } while (0)

}  // namespace

struct gmpnp_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, size = 1, device = 0;
};

struct gmpnp_group {
  std::vector<gmpnp_solver*> dom;       // local handles, ascending rank
  gmpnp_comm* comm = nullptr;           // RCCL transport (exactly one local handle) or nullptr (all ranks in this process)
  std::vector<hipStream_t> own_stream;  // in-process mode: the handles' own streams, given back at destroy
  std::vector<std::vector<int>> peer_slot;  // in-process mode: peer_slot[d][j] = index of d in the neighbour list of d's neighbour j
  int last_iters = 0;                   // BiCGStab iterations of the previous solve (identical on every rank): sizes the first burst
  // ... and those of the previous Newton solve BY NEWTON ITERATION (the k-th linear solve of a time step takes within an iteration
  // or two of what the k-th of the step before took — 85 / 65 / 53 / 44 ... — while consecutive solves differ by tens): the first
  // burst of a solve is sized by it, so that most solves end inside their first burst (a burst boundary is a device-to-host copy
  // and a stream synchronisation: 30 us of idle GPU; an iteration launched behind the end of a solve costs 14 us)
  int iters_by_newton[16] = {0};
  // Coarse operator of the two-level preconditioner: rebuilt (Galerkin product, one all-reduce, 72 x 72 inverse: 180 us in the
  // stream) for the first Newton iteration of a solve and every third one after it; in between the solves run with the
  // inverse they have — any coarse operator gives a valid right preconditioner (the single-GPU solver does the same with a
  // side stream).  A solve that needs 25 % more iterations than the last one with a fresh inverse forces a rebuild
  // (from the zero state the Jacobian of the second Newton iteration is far from the first one's: 160 instead of 73 iterations).
  // Every figure here is identical on all ranks, so all ranks decide alike.
  int coarse_age = 1 << 20, coarse_fresh_iters = 0; bool coarse_slow = false;
  // peer-mailbox transport (gmpnp_group_peer_begin / _connect): one k_peer_exchange launch per collective, no library, no host step
  bool peer = false, peer_connected = false;
  unsigned* peer_counter = nullptr;                        // arrival counter of k_dist_reduce_exchange (device)
  // exchange as the prologue of the next half-iteration's launch (k_half_a_x / k_half_b_x): possible when the launch WITH its exchange
  // workgroups is resident at once; exchange_form 0 = use it when possible, 1 = separate exchange launches (gmpnp_group_set_exchange_form)
  bool prologue_ok = false; int exchange_form = 0;
  size_t ll_red_off = 0, ll_halo_off = 0;   // flagged-word areas of the mailbox (gmpnp_dist_kernels.h, XchArgs)
  unsigned llseq = 0;                        // their sequence number (same on every rank: one per k_half_*_x launch)
  unsigned char* box = nullptr; size_t box_bytes = 0;   // own mailbox (uncached device memory)
  void* peer_map[kPeerMax] = {};                          // the other ranks' mailboxes as mapped here (IPC)
  PeerArgs pa{};
  int32_t* h_peer_err = nullptr;                          // pinned
  // caller-provided transport (gmpnp_group_create_hosted): collectives staged through pinned host memory
  bool hosted = false; gmpnp_host_transport_t host{};
  double* h_stage = nullptr; size_t h_stage_n = 0;   // pinned: [send | recv] or the all-reduce buffer
  std::vector<int64_t> off_s, cnt_s, off_r, cnt_r;
};

namespace {

// ---- peer-mailbox transport: one launch = all-reduce of `n_red` doubles (in place) and / or the ghost rows (`per` doubles a node) ----
int peer_exchange(gmpnp_group* g, double* red, int n_red, size_t per) {
  gmpnp_solver* s = g->dom[0];
  if (!g->peer_connected) return fail(GMPNP_ERR_INVALID, "peer transport: gmpnp_group_peer_connect has not been called");
  if (n_red > g->pa.red_cap) return fail(GMPNP_ERR_INVALID, "peer transport: all-reduce larger than the mailbox slot");
  if ((int)per > g->pa.wmax) return fail(GMPNP_ERR_INVALID, "peer transport: ghost rows wider than the mailbox unit");
  if (*g->h_peer_err) return fail(GMPNP_ERR_HIP, "peer transport: a rank's flag did not arrive within 5 s (rank gone, or its process ended with an error)");
  g->pa.seq++;
  hipLaunchKernelGGL(k_peer_exchange, dim3(1), dim3(1024), 0, s->stream, g->pa, (const double*)red, n_red, red,
                     (const double*)s->sendbuf.p, s->recvbuf.p, (int)per);
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

// ---- flagged-word exchange (exchange-prologue launches, gmpnp_dist_kernels.h): arguments of the next one, and what its readers poll ----
XchArgs make_xch_args(gmpnp_group* g, int phase, int par, int nout, const VecList& v, int nvec) {
  gmpnp_solver* s = g->dom[0];
  XchArgs x{};
  const PeerArgs& a = g->pa;
  for (int q = 0; q < kPeerMax; ++q) x.box[q] = a.box[q];
  x.me = a.me; x.size = a.size; x.seq = ++g->llseq; x.ll_red_off = g->ll_red_off; x.ll_halo_off = g->ll_halo_off; x.red_cap = a.red_cap;
  x.n_nb = a.n_nb;
  for (int j = 0; j < a.n_nb; ++j) { x.nb_rank[j] = a.nb_rank[j]; x.peer_recv_ptr[j] = a.peer_recv_ptr[j]; }
  for (int j = 0; j <= a.n_nb; ++j) x.send_ptr[j] = a.send_ptr[j];
  x.phase = phase; x.par = par; x.nout = nout; x.nvec = nvec;
  x.vecs = v; x.send_nodes = s->send_nodes.p;
  return x;
}
Ctx xch_ctx(gmpnp_group* g, const XchArgs& x) {
  gmpnp_solver* s = g->dom[0];
  Ctx cc = s->c;
  cc.xseq = x.seq; cc.xsize = x.size; cc.xcap = x.red_cap; cc.tile_cols_x = s->tile_cols_x.p;
  cc.xll_red = reinterpret_cast<const unsigned long long*>(g->box + g->ll_red_off);
  cc.xll_halo = reinterpret_cast<const unsigned long long*>(g->box + g->ll_halo_off);
  return cc;
}

// ---- collectives over the local handles ------------------------------------------------------------------------------------
template <class F>
int group_allreduce(gmpnp_group* g, F buf_of, int n) {
  if (g->peer) return peer_exchange(g, buf_of(g->dom[0]), n, 0);
  if (g->hosted) {
    gmpnp_solver* s = g->dom[0];
    if ((size_t)n > g->h_stage_n) return fail(GMPNP_ERR_INVALID, "hosted transport: staging buffer too small");
    HIP_TRY(hipMemcpyAsync(g->h_stage, buf_of(s), (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (g->host.allreduce(g->host.user, g->h_stage, n) != 0) return fail(GMPNP_ERR_HIP, "hosted transport: allreduce callback failed");
    HIP_TRY(hipMemcpyAsync(buf_of(s), g->h_stage, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));   // the staging buffer is reused by the next collective
    return GMPNP_OK;
  }
  if (g->comm) {
    RcclApi* api = rccl_api(nullptr);
    gmpnp_solver* s = g->dom[0];
    NCCL_TRY(api, api->AllReduce(buf_of(s), buf_of(s), (size_t)n, ncclDouble, ncclSum, g->comm->comm, s->stream));
    return GMPNP_OK;
  }
  PtrList pl{};
  for (size_t d = 0; d < g->dom.size(); ++d) pl.p[d] = buf_of(g->dom[d]);
  hipLaunchKernelGGL(k_local_allreduce, dim3(grid_for(n, 256)), dim3(256), 0, g->dom[0]->stream, pl, (int)g->dom.size(), n);
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

// Ghost rows of up to four nodal arrays (`width` doubles per node) from their owners: pack, one message per neighbour, unpack.
// The messages themselves: what every rank packed into its send buffer (`per` doubles per node) travels to the neighbours'
// receive buffers — RCCL, host-staged callbacks, or device copies between the handles of one process.
int group_transfer(gmpnp_group* g, size_t per) {
  if (g->peer) return peer_exchange(g, nullptr, 0, per);
  if (g->hosted) {
    gmpnp_solver* s = g->dom[0];
    const size_t nb = s->nb_rank.size();
    if (nb) {
      const size_t ns = (size_t)s->send_ptr.back() * per, nr = (size_t)s->recv_ptr.back() * per;
      if (ns + nr > g->h_stage_n) return fail(GMPNP_ERR_INVALID, "hosted transport: staging buffer too small");
      double* hs = g->h_stage; double* hr = g->h_stage + ns;
      g->off_s.resize(nb); g->cnt_s.resize(nb); g->off_r.resize(nb); g->cnt_r.resize(nb);
      for (size_t j = 0; j < nb; ++j) {
        g->off_s[j] = (int64_t)s->send_ptr[j] * per; g->cnt_s[j] = (int64_t)(s->send_ptr[j + 1] - s->send_ptr[j]) * per;
        g->off_r[j] = (int64_t)s->recv_ptr[j] * per; g->cnt_r[j] = (int64_t)(s->recv_ptr[j + 1] - s->recv_ptr[j]) * per;
      }
      HIP_TRY(hipMemcpyAsync(hs, s->sendbuf.p, ns * sizeof(double), hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
      if (g->host.exchange(g->host.user, (int32_t)nb, s->nb_rank.data(), g->off_s.data(), g->cnt_s.data(), hs, g->off_r.data(), g->cnt_r.data(), hr) != 0)
        return fail(GMPNP_ERR_HIP, "hosted transport: exchange callback failed");
      HIP_TRY(hipMemcpyAsync(s->recvbuf.p, hr, nr * sizeof(double), hipMemcpyHostToDevice, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
    }
  } else if (g->comm) {
    RcclApi* api = rccl_api(nullptr);
    gmpnp_solver* s = g->dom[0];
    if (!s->nb_rank.empty()) {
      NCCL_TRY(api, api->GroupStart());
      for (size_t j = 0; j < s->nb_rank.size(); ++j) {
        const size_t ns = (size_t)(s->send_ptr[j + 1] - s->send_ptr[j]) * per, nr = (size_t)(s->recv_ptr[j + 1] - s->recv_ptr[j]) * per;
        if (ns) NCCL_TRY(api, api->Send(s->sendbuf.p + (size_t)s->send_ptr[j] * per, ns, ncclDouble, s->nb_rank[j], g->comm->comm, s->stream));
        if (nr) NCCL_TRY(api, api->Recv(s->recvbuf.p + (size_t)s->recv_ptr[j] * per, nr, ncclDouble, s->nb_rank[j], g->comm->comm, s->stream));
      }
      NCCL_TRY(api, api->GroupEnd());
    }
  } else {
    for (size_t d = 0; d < g->dom.size(); ++d) {
      gmpnp_solver* s = g->dom[d];
      for (size_t j = 0; j < s->nb_rank.size(); ++j) {
        gmpnp_solver* q = g->dom[s->nb_rank[j]];
        const int jj = g->peer_slot[d][j];
        const size_t ns = (size_t)(s->send_ptr[j + 1] - s->send_ptr[j]) * per;
        if (ns) HIP_TRY(hipMemcpyAsync(q->recvbuf.p + (size_t)q->recv_ptr[jj] * per, s->sendbuf.p + (size_t)s->send_ptr[j] * per,
                                       ns * sizeof(double), hipMemcpyDeviceToDevice, g->dom[0]->stream));
      }
    }
  }
  return GMPNP_OK;
}

// the all-reduce AND the ghost rows of a BiCGStab half-iteration: one launch on the peer transport, two collectives otherwise
template <class F>
int group_reduce_transfer(gmpnp_group* g, F buf_of, int n, size_t per) {
  if (g->peer) return peer_exchange(g, buf_of(g->dom[0]), n, per);
  int r = group_allreduce(g, buf_of, n); if (r) return r;
  return group_transfer(g, per);
}

template <class F>
int group_exchange(gmpnp_group* g, int width, int nvec, F vecs_of) {
  for (gmpnp_solver* s : g->dom) {
    const int nsn = s->send_ptr.empty() ? 0 : s->send_ptr.back();
    if (nsn == 0) continue;
    VecListW w = vecs_of(s); VecList src{};
    for (int v = 0; v < 4; ++v) src.p[v] = w.p[v];
    hipLaunchKernelGGL(k_halo_pack, dim3(grid_for(nsn * nvec * width, 256)), dim3(256), 0, s->stream, src, nvec, width,
                       (const int32_t*)s->send_nodes.p, nsn, s->sendbuf.p);
  }
  HIP_TRY(hipGetLastError());
  { int rt = group_transfer(g, (size_t)nvec * width); if (rt) return rt; }
  for (gmpnp_solver* s : g->dom) {
    const int nrn = s->recv_ptr.empty() ? 0 : s->recv_ptr.back();
    if (nrn == 0) continue;
    hipLaunchKernelGGL(k_halo_unpack, dim3(grid_for(nrn * nvec * width, 256)), dim3(256), 0, s->stream, vecs_of(s), nvec, width,
                       (const int32_t*)s->recv_nodes.p, nrn, (const double*)s->recvbuf.p);
  }
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

// ---- residual: ||b||_2 over all ranks' owned rows and the OR of the status bits -----------------------------------------------
template <int DIM, int NF>
int group_residual(gmpnp_group* g, double* norm, int* flags) {
  for (gmpnp_solver* s : g->dom) {
    int rc = launch_element<DIM, NF>(s, true); if (rc) return rc;
    rc = launch_res_gather<DIM, NF>(s); if (rc) return rc;
    hipLaunchKernelGGL(k_norm_reduce, dim3(1), dim3(256), 0, s->stream, (const double*)s->c.part_f, s->n_resblocks,
                       (const int32_t*)s->status.p, s->red_norm.p);
  }
  int rc = group_allreduce(g, [](gmpnp_solver* s) { return s->red_norm.p; }, 5); if (rc) return rc;
  gmpnp_solver* s0 = g->dom[0];
  HIP_TRY(hipMemcpyAsync(s0->h_red, s0->red_norm.p, 5 * sizeof(double), hipMemcpyDeviceToHost, s0->stream));
  HIP_TRY(hipStreamSynchronize(s0->stream));
  if (g->peer && *g->h_peer_err) return fail(GMPNP_ERR_HIP, "peer transport: a rank's flag did not arrive within 5 s");
  *norm = std::sqrt(s0->h_red[0]);
  int f = 0;
  for (int b = 0; b < 4; ++b) if (s0->h_red[1 + b] > 0.0) f |= 1 << b;
  *flags = f;
  return GMPNP_OK;
}

// ---- preconditioner of the partitioned operator: node-block Jacobi (ghost blocks from their owners) + GLOBAL slab coarse space ----
template <int NF>
int group_setup(gmpnp_group* g, int mode, bool rebuild_coarse = true) {
  const int use_coarse = (mode == GMPNP_LINEAR_BICGSTAB_TWOLEVEL) ? 1 : 0;
  for (gmpnp_solver* s : g->dom) {
    s->c.use_coarse = use_coarse;
    hipLaunchKernelGGL((k_block_inverse<NF>), dim3(grid_for(s->t.nv, 4)), dim3(64), 0, s->stream, s->c);
  }
  // As = J Dinv needs the owners' inverse blocks at the ghost COLUMNS (the local ghost rows are identity rows)
  int rc = group_exchange(g, NF * NF, 1, [](gmpnp_solver* s) { VecListW w{}; w.p[0] = s->Dinv.p; return w; }); if (rc) return rc;
  for (gmpnp_solver* s : g->dom)
    hipLaunchKernelGGL((k_scale_columns<NF>), dim3(grid_for(s->c.n_work * kWave, kVecBlock)), dim3(kVecBlock), 0, s->stream, s->c);
  if (use_coarse && rebuild_coarse) {
    for (gmpnp_solver* s : g->dom) {
      const int n = s->ncoarse;
      hipLaunchKernelGGL((k_coarse_rows<NF>), dim3(s->t.nslices), dim3(64), 0, s->stream, s->c);
      hipLaunchKernelGGL((k_coarse_sum<NF>), dim3(s->t.nagg * s->c.coarse_chunks), dim3(kVecBlock), 0, s->stream, s->c);
      hipLaunchKernelGGL(k_coarse_reduce, dim3(grid_for(n * n, kVecBlock)), dim3(kVecBlock), 0, s->stream, s->c);
      hipLaunchKernelGGL(k_zero_foreign_rows, dim3(grid_for(n * n, 256)), dim3(256), 0, s->stream, s->Ac.p, n, s->t.own_agg0 * NF, s->t.own_agg1 * NF);
    }
    const int n = g->dom[0]->ncoarse;
    rc = group_allreduce(g, [](gmpnp_solver* s) { return s->Ac.p; }, n * n); if (rc) return rc;   // ONE all-reduce per set-up
    for (gmpnp_solver* s : g->dom)
      hipLaunchKernelGGL((k_coarse_invert<NF>), dim3(1), dim3(512), coarse_lds_bytes(n, NF), s->stream, s->c);
  }
  HIP_TRY(hipGetLastError());
  for (gmpnp_solver* s : g->dom) { s->precond_valid = true; s->precond_mode = mode; }
  return GMPNP_OK;
}

// ---- BiCGStab across the ranks: rhs in kr (owned rows; k_res_gather left it there), ||rhs|| = bnorm (global); leaves y in ky ----
// random_shadow: the shadow vector of this pass is each handle's krand (filled by the caller) and (rhat, r_0) = shadow_rho0
template <int NF>
int group_krylov(gmpnp_group* g, int mode, double bnorm, double rtol, double atol, int maxit, gmpnp_linear_stats_t* st, bool sized_by_previous = true,
                 bool random_shadow = false, double shadow_rho0 = 0.0, int predicted = 0) {
  const int use_coarse = (mode == GMPNP_LINEAR_BICGSTAB_TWOLEVEL) ? 1 : 0;
  const int n = g->dom[0]->ncoarse;
  KrylovScalars init{};
  init.rho[0] = init.rho[1] = random_shadow ? shadow_rho0 : bnorm * bnorm; init.alpha = 1.0;
  init.tol = std::max(rtol * bnorm, atol); init.rr = bnorm * bnorm; init.max_iters = maxit; init.rr0 = bnorm * bnorm;
  if (!(bnorm > 0.0)) init.done = 1;
  for (gmpnp_solver* s : g->dom) {
    s->c.use_coarse = use_coarse;
    hipLaunchKernelGGL((k_krylov_init<NF>), dim3(s->t.own_ntiles), dim3(kVecBlock), 0, s->stream, s->c,
                       random_shadow ? (const double*)s->krand.p : (const double*)nullptr, init, s->cpart_v1.p);
    if (use_coarse) hipLaunchKernelGGL(k_dist_reduce, dim3(n), dim3(256), 0, s->stream, s->c, 0, 0, s->red_i.p);
  }
  HIP_TRY(hipGetLastError());
  int rc;
  if (use_coarse) { rc = group_allreduce(g, [](gmpnp_solver* s) { return s->red_i.p; }, n); if (rc) return rc; }
  rc = group_exchange(g, NF, 1, [](gmpnp_solver* s) { VecListW w{}; w.p[0] = s->kr.p; return w; }); if (rc) return rc;   // p_0 = r_0 at the ghost columns
  const dim3 cg(std::max(1, g->dom[0]->t.nagg));
  const dim3 cg_peer = cg;
  KrylovScalars res = init;
  int k = 0;
  // Per half-iteration and rank THREE launches: [coarse kernel + unpacking of the ghost rows received last], tile kernel,
  // [per-rank sums + packing of the ghost rows to send]; then the all-reduce and the grouped send/recv.
  // Peer transport: the sums, the ghost rows and their exchange are ONE launch per half-iteration (k_dist_reduce_exchange), the
  // received rows are in place when it ends: coarse kernel, tile kernel, exchange — 6 launches per iteration (4 where the coarse workgroups ride inside the tile launch), no library call.
  auto iteration_peer = [&]() -> int {
    const int par = k & 1;
    gmpnp_solver* s = g->dom[0];
    if (!g->peer_connected) return fail(GMPNP_ERR_INVALID, "peer transport: gmpnp_group_peer_connect has not been called");
    if (*g->h_peer_err) return fail(GMPNP_ERR_HIP, "peer transport: a rank's flag did not arrive within 5 s");
    const int nsn = s->send_ptr.empty() ? 0 : s->send_ptr.back();
    // (the received rows are in place and the sums all-reduced when the previous launch ends, so the coarse workgroups can
    // ride in front of the tile workgroups as on one GPU wherever the whole launch is resident: 4 launches per iteration)
    const dim3 fg(s->t.nagg + s->t.own_ntiles);
    if (s->fused_half && g->prologue_ok && g->exchange_form != 1) {
      // TWO launches per iteration: the exchange of a launch's sums and boundary rows rides in front of the NEXT launch's coarse
      // workgroups (gmpnp_dist_kernels.h, "exchange as the prologue").  A(0) needs nothing exchanged (the start-up collectives did it).
      auto xargs = [&](int phase, int nout, const VecList& v, int nvec) { XchArgs x = make_xch_args(g, phase, par, nout, v, nvec); x.nsn = nsn; x.nx = xch_workgroups(nout, nsn, nvec, NF); return x; };
      auto xctx = [&](const XchArgs& x) { return xch_ctx(g, x); };
      if (k == 0) hipLaunchKernelGGL((k_half_a<NF>), fg, dim3(kKrylovThreads), 0, s->stream, s->c, k, (unsigned)(++s->fused_seq));
      else {   // prologue: what B(k-1) left (sums of phase 2, rows of s and t)
        VecList vb{}; vb.p[0] = s->ks.p; vb.p[1] = s->kt.p;
        const XchArgs x = xargs(2, 4 + n, vb, 2);
        const Ctx cc = xctx(x);
        hipLaunchKernelGGL((k_half_a_x<NF>), dim3(x.nx + fg.x), dim3(kKrylovThreads), 0, s->stream, cc, k, (unsigned)(++s->fused_seq), x);
      }
      {        // prologue: what A(k) left (sums of phase 1, rows of r, v, p)
        VecList va{}; va.p[0] = s->kr.p; va.p[1] = s->c.kv[par]; va.p[2] = s->c.kp[par];
        const XchArgs x = xargs(1, 2 + 3 * n, va, 3);
        const Ctx cc = xctx(x);
        hipLaunchKernelGGL((k_half_b_x<NF>), dim3(x.nx + fg.x), dim3(kKrylovThreads), 0, s->stream, cc, k, (unsigned)(++s->fused_seq), x);
      }
      ++k;
      HIP_TRY(hipGetLastError());
      return GMPNP_OK;
    }
    if (s->fused_half) hipLaunchKernelGGL((k_half_a<NF>), fg, dim3(kKrylovThreads), 0, s->stream, s->c, k, (unsigned)(++s->fused_seq));
    else {
      hipLaunchKernelGGL((k_coarse_a<NF>), cg_peer, dim3(kCoarseThreads), 0, s->stream, s->c, k);
      hipLaunchKernelGGL((k_bicg_a<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, k);
    }
    VecListW va{}; va.p[0] = s->kr.p; va.p[1] = s->c.kv[par]; va.p[2] = s->c.kp[par];
    g->pa.seq++;
    hipLaunchKernelGGL(k_dist_reduce_exchange, dim3(2 + 3 * n + grid_for(nsn * 3 * NF, 256)), dim3(256), 0, s->stream, s->c, 1, par, s->red_a.p, 2 + 3 * n,
                       va, 3, NF, (const int32_t*)s->send_nodes.p, nsn, (const int32_t*)s->recv_nodes.p, g->pa, g->peer_counter);
    if (s->fused_half) hipLaunchKernelGGL((k_half_b<NF>), fg, dim3(kKrylovThreads), 0, s->stream, s->c, k, (unsigned)(++s->fused_seq));
    else {
      hipLaunchKernelGGL((k_coarse_b<NF>), cg_peer, dim3(kCoarseThreads), 0, s->stream, s->c, k);
      hipLaunchKernelGGL((k_bicg_b<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, k);
    }
    VecListW vb{}; vb.p[0] = s->ks.p; vb.p[1] = s->kt.p;
    g->pa.seq++;
    hipLaunchKernelGGL(k_dist_reduce_exchange, dim3(4 + n + grid_for(nsn * 2 * NF, 256)), dim3(256), 0, s->stream, s->c, 2, par, s->red_b.p, 4 + n,
                       vb, 2, NF, (const int32_t*)s->send_nodes.p, nsn, (const int32_t*)s->recv_nodes.p, g->pa, g->peer_counter);
    ++k;
    HIP_TRY(hipGetLastError());
    return GMPNP_OK;
  };
  auto iteration = [&]() -> int {
    if (g->peer) return iteration_peer();
    const int par = k & 1;
    for (gmpnp_solver* s : g->dom) {
      const int nrn = s->recv_ptr.empty() ? 0 : s->recv_ptr.back(), nsn = s->send_ptr.empty() ? 0 : s->send_ptr.back();
      VecListW ub{}; ub.p[0] = s->ks.p; ub.p[1] = s->kt.p;    // what B(k-1) sent (nothing pending before the first iteration)
      const int un = k > 0 ? nrn : 0;
      hipLaunchKernelGGL((k_coarse_a_unpack<NF>), dim3(cg.x + grid_for(un * 2 * NF, kCoarseThreads)), dim3(kCoarseThreads), 0, s->stream, s->c, k, ub, 2,
                         (const int32_t*)s->recv_nodes.p, un, (const double*)s->recvbuf.p);
      hipLaunchKernelGGL((k_bicg_a<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, k);
      VecList pa{}; pa.p[0] = s->kr.p; pa.p[1] = s->c.kv[par]; pa.p[2] = s->c.kp[par];
      hipLaunchKernelGGL(k_dist_reduce_pack, dim3(2 + 3 * n + grid_for(nsn * 3 * NF, 256)), dim3(256), 0, s->stream, s->c, 1, par, s->red_a.p, 2 + 3 * n, pa, 3,
                         (const int32_t*)s->send_nodes.p, nsn, s->sendbuf.p, NF);
    }
    int r = group_reduce_transfer(g, [](gmpnp_solver* s) { return s->red_a.p; }, 2 + 3 * n, (size_t)3 * NF); if (r) return r;
    for (gmpnp_solver* s : g->dom) {
      const int nrn = s->recv_ptr.empty() ? 0 : s->recv_ptr.back(), nsn = s->send_ptr.empty() ? 0 : s->send_ptr.back();
      VecListW ua{}; ua.p[0] = s->kr.p; ua.p[1] = s->c.kv[par]; ua.p[2] = s->c.kp[par];
      hipLaunchKernelGGL((k_coarse_b_unpack<NF>), dim3(cg.x + grid_for(nrn * 3 * NF, kCoarseThreads)), dim3(kCoarseThreads), 0, s->stream, s->c, k, ua, 3,
                         (const int32_t*)s->recv_nodes.p, nrn, (const double*)s->recvbuf.p);
      hipLaunchKernelGGL((k_bicg_b<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, k);
      VecList pb{}; pb.p[0] = s->ks.p; pb.p[1] = s->kt.p;
      hipLaunchKernelGGL(k_dist_reduce_pack, dim3(4 + n + grid_for(nsn * 2 * NF, 256)), dim3(256), 0, s->stream, s->c, 2, par, s->red_b.p, 4 + n, pb, 2,
                         (const int32_t*)s->send_nodes.p, nsn, s->sendbuf.p, NF);
    }
    r = group_reduce_transfer(g, [](gmpnp_solver* s) { return s->red_b.p; }, 4 + n, (size_t)2 * NF); if (r) return r;
    ++k;
    HIP_TRY(hipGetLastError());
    return GMPNP_OK;
  };
  if (!res.done) {
    // Bursts: every rank launches the SAME number of iterations (the schedule depends only on the previous solve's count and
    // on `done`, both identical on all ranks), then reads the device's verdict.  Iterations launched behind the end of the
    // solve exit at their first instruction; their collectives still pair up.
    int burst = predicted > 0 ? predicted + 1 : (sized_by_previous ? std::max(2, (7 * g->last_iters) / 8) : 4);
    gmpnp_solver* s0 = g->dom[0];
    while (true) {
      for (int it = 0; it < burst; ++it) { rc = iteration(); if (rc) return rc; }
      HIP_TRY(hipMemcpyAsync(&s0->h_scal[0], s0->scal.p, sizeof(KrylovScalars), hipMemcpyDeviceToHost, s0->stream));
      for (gmpnp_solver* s : g->dom) HIP_TRY(hipStreamSynchronize(s->stream));
      if (g->peer && *g->h_peer_err) return fail(GMPNP_ERR_HIP, "peer transport: a rank's flag did not arrive within 5 s");
      res = s0->h_scal[0];
      if (res.done) break;
      if (k > maxit + 8) break;
      burst = 4;
    }
  }
  g->last_iters = res.iters;
  for (gmpnp_solver* s : g->dom) s->last_done = res.done;
  if (st) { st->iterations = res.iters; st->converged = (res.done == 1); st->residual_norm = std::sqrt(res.rr); st->rhs_norm = bnorm; }
  if (res.done != 1) {
    char buf[200];
    snprintf(buf, sizeof buf, "partitioned BiCGStab stopped without convergence (code %d) after %d iterations, ||r|| = %.3e, ||b|| = %.3e",
             res.done, res.iters, std::sqrt(res.rr), bnorm);
    return fail(GMPNP_ERR_LINEAR, buf);
  }
  return GMPNP_OK;
}

// x = Dinv (I + P Aci P^T) y on the owned rows, ghost rows from their owners, then u -= omega x on every local row
template <int NF>
int group_update(gmpnp_group* g, int mode, double omega, bool add_to_start) {
  const int use_coarse = (mode == GMPNP_LINEAR_BICGSTAB_TWOLEVEL) ? 1 : 0;
  const int n = g->dom[0]->ncoarse;
  if (use_coarse) {
    for (gmpnp_solver* s : g->dom) {
      hipLaunchKernelGGL((k_restrict<NF>), dim3(s->t.own_ntiles), dim3(kVecBlock), 0, s->stream, s->c, (const double*)s->ky.p, s->cpart_v0.p);
      hipLaunchKernelGGL(k_dist_reduce, dim3(n), dim3(256), 0, s->stream, s->c, 3, 0, s->red_i.p);
    }
    int rc = group_allreduce(g, [](gmpnp_solver* s) { return s->red_i.p; }, n); if (rc) return rc;
  }
  for (gmpnp_solver* s : g->dom)
    hipLaunchKernelGGL((k_minv_apply<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, (const double*)s->ky.p,
                       (const double*)s->cpart_v0.p, s->kx.p, add_to_start ? 1.0 : 0.0, 1.0, NewtonUpdate{nullptr, nullptr, 0.0, 0.0, 0.0},
                       (const double*)s->red_i.p);
  int rc = group_exchange(g, NF, 1, [](gmpnp_solver* s) { VecListW w{}; w.p[0] = s->kx.p; return w; }); if (rc) return rc;
  for (gmpnp_solver* s : g->dom)
    hipLaunchKernelGGL(k_axpy, dim3(grid_for(s->ndof, 256)), dim3(256), 0, s->stream, s->u.p, (const double*)s->kx.p, -omega, (int)s->ndof);
  HIP_TRY(hipGetLastError());
  return GMPNP_OK;
}

template <int DIM, int NF>
int group_newton(gmpnp_group* g, const gmpnp_newton_options_t& o, gmpnp_newton_stats_t& st) {
  const double t0 = now_ms();
  for (gmpnp_solver* s : g->dom) HIP_TRY(hipMemsetAsync(s->status.p, 0, sizeof(int32_t), s->stream));
  // ghost values of u and u_n are the owners' values from here on (the caller's scatter normally made them so already)
  int rc = group_exchange(g, NF, 2, [](gmpnp_solver* s) { VecListW w{}; w.p[0] = s->u.p; w.p[1] = s->un.p; return w; }); if (rc) return rc;
  double r = 0.0; int flags = 0;
  rc = group_residual<DIM, NF>(g, &r, &flags); if (rc) return rc;
  if (flags & 1) { st.steric_excursion = 1; if (g->dom[0]->strict_steric) return fail(GMPNP_ERR_NUMERIC, status_message(flags)); }
  if (!(r == r)) return fail(GMPNP_ERR_NUMERIC, "residual is NaN before the first Newton iteration");
  const double r0 = r;
  st.residuals[0] = r; st.n_residuals = 1;
  auto conv = [&](double res) { const double rel = res / r0; return rel < o.relative_tolerance || res < o.absolute_tolerance; };
  bool done = conv(r);
  while (!done && st.iterations < o.maximum_iterations) {
    for (gmpnp_solver* s : g->dom) { rc = launch_jac_gather<DIM, NF>(s); if (rc) return rc; s->jacobian_valid = true; }
    const bool two_level = o.linear_solver == GMPNP_LINEAR_BICGSTAB_TWOLEVEL;
    // (a solve that starts from a state set from outside — the zero state of time step 0 — rebuilds every time: its Jacobians differ
    // too much, 203 instead of 56 iterations with the first iteration's inverse in the second)
    const bool rebuild = two_level && (st.iterations == 0 || g->coarse_age >= 2 || g->coarse_slow || g->dom[0]->state_jumped);
    rc = group_setup<NF>(g, o.linear_solver, rebuild); if (rc) return rc;
    gmpnp_linear_stats_t ls{};
    // Warm start, as in the single-GPU Newton (gmpnp_api.hip): with the damped update consecutive corrections satisfy
    // dx_{k+1} = (1 - w) dx_k + O(|dx_k|^2); x0 = (1-w) dx_k [+ (1-w)^2 (dx_k - (1-w) dx_{k-1})] is accepted when it removes at
    // least half of the residual (one SpMV, three all-reduced dot products, a decision identical on every rank), and
    // BiCGStab then only has to remove b - J x0, to the SAME absolute target.
    const double tol_abs = std::max(o.krylov_relative_tolerance * r, o.krylov_absolute_tolerance);
    const double q = 1.0 - o.relaxation_parameter;
    bool warm = false; double rstart = r;
    if (g->dom[0]->warm_start && st.iterations > 0 && q != 0.0 && r > 0.0) {
      const double wa = (g->dom[0]->warm_start > 1 && st.iterations > 1) ? q + q * q : q, wb = (g->dom[0]->warm_start > 1 && st.iterations > 1) ? -q * q * q : 0.0;
      for (gmpnp_solver* s : g->dom) {
        const int n = s->ndof;
        hipLaunchKernelGGL(k_warm_start, dim3(grid_for(n, 256)), dim3(256), 0, s->stream, s->kx.p, s->kxp.p, wa, wb, n);
        hipLaunchKernelGGL((k_spmv_plain<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, (const double*)s->kx.p, s->kt.p);
        hipLaunchKernelGGL(k_dots3, dim3(s->n_resblocks), dim3(kVecBlock), 0, s->stream, (const double*)s->kt.p, (const double*)s->kb.p, s->c.part_f, n,
                           s->n_resblocks, s->t.own_node0 * NF, s->t.own_node1 * NF);
        hipLaunchKernelGGL(k_dots3_reduce, dim3(1), dim3(256), 0, s->stream, (const double*)s->c.part_f, s->n_resblocks, s->red_norm.p);
      }
      rc = group_allreduce(g, [](gmpnp_solver* s) { return s->red_norm.p; }, 3); if (rc) return rc;
      gmpnp_solver* s0 = g->dom[0];
      HIP_TRY(hipMemcpyAsync(s0->h_red, s0->red_norm.p, 3 * sizeof(double), hipMemcpyDeviceToHost, s0->stream));
      HIP_TRY(hipStreamSynchronize(s0->stream));
      const double wbd = s0->h_red[0], ww = s0->h_red[1], bb = s0->h_red[2], rn2 = bb - 2.0 * wbd + ww;
      if (rn2 == rn2 && rn2 >= 0.0 && rn2 < 0.25 * bb) {
        for (gmpnp_solver* s : g->dom)
          hipLaunchKernelGGL(k_start_residual, dim3(grid_for(s->ndof, 256)), dim3(256), 0, s->stream, s->kr.p, (const double*)s->kb.p, (const double*)s->kt.p, (int)s->ndof);
        warm = true; rstart = std::sqrt(rn2);
      }
    }
    int kry_total = 0;
    if (warm && rstart <= tol_abs) { ls.converged = 1; ls.residual_norm = rstart; for (gmpnp_solver* s : g->dom) HIP_TRY(hipMemsetAsync(s->ky.p, 0, s->ndof * sizeof(double), s->stream)); }
    else {
      // BiCGStab can break down, or spike past 1e5 times its starting residual, on one (right-hand side, shadow vector) pair
      // and run smoothly on another: such a pass is thrown away and repeated with a pseudo-random shadow vector — the
      // second time also without the predicted start — as the single-GPU solver does (gmpnp_api.hip, linear_solve).  The
      // verdict comes from all-reduced sums, so every rank takes the same branch.
      bool random_shadow = false; double rho0 = 0.0;
      for (int attempt = 0;; ++attempt) {
        const int hist = (attempt == 0 && !g->dom[0]->state_jumped && st.iterations < 16) ? g->iters_by_newton[st.iterations] : 0;
        rc = group_krylov<NF>(g, o.linear_solver, rstart, warm ? 0.0 : o.krylov_relative_tolerance, warm ? tol_abs : o.krylov_absolute_tolerance,
                              o.krylov_maximum_iterations, &ls, st.iterations > 0 && attempt == 0, random_shadow, rho0, hist);
        if (attempt == 0 && rc == GMPNP_OK && st.iterations < 16) g->iters_by_newton[st.iterations] = ls.iterations;
        kry_total += ls.iterations;
        if (rc != GMPNP_ERR_LINEAR || attempt >= 4 || g->dom[0]->last_done != 3) break;
        if (warm && attempt >= 1) { warm = false; rstart = r; }
        for (gmpnp_solver* s : g->dom) {
          const int nd = s->ndof;
          HIP_TRY(hipMemsetAsync(s->status.p, 0, sizeof(int32_t), s->stream));
          if (warm) {   // kr = b - J x0 again (kt was a work vector of the lost pass)
            hipLaunchKernelGGL((k_spmv_plain<NF>), dim3(s->t.own_ntiles), dim3(kKrylovThreads), 0, s->stream, s->c, (const double*)s->kx.p, s->kt.p);
            hipLaunchKernelGGL(k_start_residual, dim3(grid_for(nd, 256)), dim3(256), 0, s->stream, s->kr.p, (const double*)s->kb.p, (const double*)s->kt.p, nd);
          } else HIP_TRY(hipMemcpyAsync(s->kr.p, s->kb.p, nd * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
          hipLaunchKernelGGL(k_fill_hash, dim3(grid_for(nd, 256)), dim3(256), 0, s->stream, s->krand.p, (unsigned)((attempt + 1) * 2654435761u), nd);
          hipLaunchKernelGGL(k_dots3, dim3(s->n_resblocks), dim3(kVecBlock), 0, s->stream, (const double*)s->krand.p, (const double*)s->kr.p, s->c.part_f, nd,
                             s->n_resblocks, s->t.own_node0 * NF, s->t.own_node1 * NF);
          hipLaunchKernelGGL(k_dots3_reduce, dim3(1), dim3(256), 0, s->stream, (const double*)s->c.part_f, s->n_resblocks, s->red_norm.p);
        }
        HIP_TRY(hipGetLastError());
        { int r2 = group_allreduce(g, [](gmpnp_solver* s) { return s->red_norm.p; }, 3); if (r2) return r2; }
        gmpnp_solver* s0 = g->dom[0];
        HIP_TRY(hipMemcpyAsync(s0->h_red, s0->red_norm.p, 3 * sizeof(double), hipMemcpyDeviceToHost, s0->stream));
        HIP_TRY(hipStreamSynchronize(s0->stream));
        rho0 = s0->h_red[0]; random_shadow = true;   // (rhat, r_0) of the new shadow vector, over all ranks' owned rows
      }
      ls.iterations = kry_total;
    }
    if (st.iterations < GMPNP_MAX_NEWTON_HISTORY) st.krylov_per_iteration[st.iterations] = ls.iterations;
    st.krylov_iterations += ls.iterations;
    if (rc) return rc;
    if (two_level) {
      if (rebuild) { g->coarse_age = 0; g->coarse_fresh_iters = ls.iterations; g->coarse_slow = false; }
      else { g->coarse_age++; g->coarse_slow = ls.iterations > g->coarse_fresh_iters + g->coarse_fresh_iters / 4 + 5; }
    }
    rc = group_update<NF>(g, o.linear_solver, o.relaxation_parameter, warm); if (rc) return rc;
    st.iterations++;
    rc = group_residual<DIM, NF>(g, &r, &flags); if (rc) return rc;
    if (flags & 1) { st.steric_excursion = 1; if (g->dom[0]->strict_steric) return fail(GMPNP_ERR_NUMERIC, status_message(flags)); }
    if (flags & 14) return fail(GMPNP_ERR_LINEAR, status_message(flags));
    if (st.n_residuals < GMPNP_MAX_NEWTON_HISTORY) st.residuals[st.n_residuals++] = r;
    if (!(r == r) || std::isinf(r)) return fail(GMPNP_ERR_NUMERIC, "residual became NaN / Inf");
    done = conv(r);
  }
  for (gmpnp_solver* s : g->dom) { s->state_jumped = false; s->x0_predicted = false; }
  st.converged = done ? 1 : 0;
  st.ms_total = now_ms() - t0;
  if (!done) return fail(GMPNP_ERR_NOT_CONVERGED, "Newton solver did not converge because maximum number of iterations reached");
  return GMPNP_OK;
}

}  // namespace

extern "C" {

int gmpnp_comm_unique_id(char id[GMPNP_COMM_ID_BYTES]) {
  static_assert(GMPNP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id) return fail(GMPNP_ERR_INVALID, "NULL argument");
  std::string why;
  RcclApi* api = rccl_api(&why);
  if (!api) return fail(GMPNP_ERR_HIP, why);
  ncclUniqueId u;
  NCCL_TRY(api, api->GetUniqueId(&u));
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return GMPNP_OK;
}

int gmpnp_comm_create(const char id[GMPNP_COMM_ID_BYTES], int32_t rank, int32_t size, int32_t device_id, gmpnp_comm** out) {
  if (!id || !out || size < 1 || rank < 0 || rank >= size) return fail(GMPNP_ERR_INVALID, "bad arguments");
  *out = nullptr;
  std::string why;
  RcclApi* api = rccl_api(&why);
  if (!api) return fail(GMPNP_ERR_HIP, why);
  HIP_TRY(hipSetDevice(device_id));
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  std::unique_ptr<gmpnp_comm> c(new gmpnp_comm);
  c->rank = rank; c->size = size; c->device = device_id;
  NCCL_TRY(api, api->CommInitRank(&c->comm, size, u, rank));
  *out = c.release();
  return GMPNP_OK;
}

// Round trip through every RCCL entry point the partitioned solve uses, on this rank alone: n doubles sent to OUR OWN rank
// and received back inside one group (RCCL pairs a send-to-self with the matching receive), then all-reduced.  A single-GPU
// box cannot host a second rank (RCCL refuses two ranks on one device), so this is how the send/receive bindings get
// exercised there.  Collective in the sense that every rank of the communicator has to call it (the all-reduce).
int gmpnp_comm_selftest(gmpnp_comm* c, int32_t n, double* max_error) {
  if (!c || n < 1 || !max_error) return fail(GMPNP_ERR_INVALID, "bad arguments");
  RcclApi* api = rccl_api(nullptr);
  if (!api) return fail(GMPNP_ERR_HIP, "RCCL not loaded");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st; HIP_TRY(hipStreamCreate(&st));
  DevBuf<double> a, b;
  HIP_TRY(a.alloc(n)); HIP_TRY(b.alloc(n));
  std::vector<double> h(n), back(n);
  for (int i = 0; i < n; ++i) h[i] = 0.25 * i - 3.0 + c->rank;
  HIP_TRY(hipMemcpyAsync(a.p, h.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
  NCCL_TRY(api, api->GroupStart());
  NCCL_TRY(api, api->Send(a.p, (size_t)n, ncclDouble, c->rank, c->comm, st));
  NCCL_TRY(api, api->Recv(b.p, (size_t)n, ncclDouble, c->rank, c->comm, st));
  NCCL_TRY(api, api->GroupEnd());
  NCCL_TRY(api, api->AllReduce(b.p, b.p, (size_t)n, ncclDouble, ncclSum, c->comm, st));
  HIP_TRY(hipMemcpyAsync(back.data(), b.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  (void)hipStreamDestroy(st);
  double err = 0.0;
  for (int i = 0; i < n; ++i) {   // sum over ranks r of (0.25 i - 3 + r)
    const double want = c->size * (0.25 * i - 3.0) + 0.5 * c->size * (c->size - 1);
    err = std::max(err, std::fabs(back[i] - want));
  }
  *max_error = err;
  return GMPNP_OK;
}

void gmpnp_comm_destroy(gmpnp_comm* c) {
  if (!c) return;
  RcclApi* api = rccl_api(nullptr);
  if (api && c->comm) { (void)hipSetDevice(c->device); (void)api->CommDestroy(c->comm); }
  delete c;
}

int gmpnp_group_create(int32_t n_local, gmpnp_solver* const* handles, gmpnp_comm* comm, gmpnp_group** out) {
  if (n_local < 1 || !handles || !out) return fail(GMPNP_ERR_INVALID, "bad arguments");
  *out = nullptr;
  std::unique_ptr<gmpnp_group> g(new gmpnp_group);
  for (int d = 0; d < n_local; ++d) {
    gmpnp_solver* s = handles[d];
    if (!s || !s->partitioned) return fail(GMPNP_ERR_INVALID, "group members must come from gmpnp_create_partition");
    g->dom.push_back(s);
  }
  gmpnp_solver* s0 = g->dom[0];
  for (gmpnp_solver* s : g->dom)
    if (s->part_size != s0->part_size || s->ncoarse != s0->ncoarse || s->nf != s0->nf || s->opts.device_id != s0->opts.device_id)
      return fail(GMPNP_ERR_INVALID, "group members disagree on partition size, coarse space, fields or device");
  if (comm) {
    if (n_local != 1) return fail(GMPNP_ERR_INVALID, "with a communicator a process drives exactly one partition handle");
    if (comm->size != s0->part_size || comm->rank != s0->part_rank) return fail(GMPNP_ERR_INVALID, "communicator rank/size differ from the partition's");
    g->comm = comm;
  } else {
    if (n_local != s0->part_size || n_local > 8) return fail(GMPNP_ERR_INVALID, "without a communicator the group must hold every rank of the partition (at most 8)");
    for (int d = 0; d < n_local; ++d) if (g->dom[d]->part_rank != d) return fail(GMPNP_ERR_INVALID, "handles must be given in rank order");
    g->peer_slot.resize(n_local);
    for (int d = 0; d < n_local; ++d) {
      gmpnp_solver* s = g->dom[d];
      for (size_t j = 0; j < s->nb_rank.size(); ++j) {
        gmpnp_solver* q = g->dom[s->nb_rank[j]];
        int jj = -1;
        for (size_t z = 0; z < q->nb_rank.size(); ++z) if (q->nb_rank[z] == d) jj = (int)z;
        if (jj < 0 || (q->recv_ptr[jj + 1] - q->recv_ptr[jj]) != (s->send_ptr[j + 1] - s->send_ptr[j]))
          return fail(GMPNP_ERR_INVALID, "halo plans of two neighbouring ranks do not match");
        g->peer_slot[d].push_back(jj);
      }
    }
    // one stream for all handles of the process: their launches and the copies between them are ordered without events
    HIP_TRY(hipSetDevice(s0->opts.device_id));
    g->own_stream.resize(n_local, nullptr);
    for (int d = 1; d < n_local; ++d) {
      HIP_TRY(hipStreamSynchronize(g->dom[d]->stream));
      g->own_stream[d] = g->dom[d]->stream; g->dom[d]->stream = s0->stream;
    }
  }
  *out = g.release();
  return GMPNP_OK;
}

int gmpnp_group_create_hosted(gmpnp_solver* handle, const gmpnp_host_transport_t* t, gmpnp_group** out) {
  if (!handle || !t || !out || !t->allreduce || !t->exchange) return fail(GMPNP_ERR_INVALID, "bad arguments");
  *out = nullptr;
  if (!handle->partitioned) return fail(GMPNP_ERR_INVALID, "group members must come from gmpnp_create_partition");
  if (t->size != handle->part_size || t->rank != handle->part_rank) return fail(GMPNP_ERR_INVALID, "transport rank/size differ from the partition's");
  std::unique_ptr<gmpnp_group> g(new gmpnp_group);
  g->dom.push_back(handle);
  g->hosted = true; g->host = *t;
  const size_t n = (size_t)handle->ncoarse;
  g->h_stage_n = std::max<size_t>({handle->sendbuf.n + handle->recvbuf.n, n * n, 2 + 3 * n, (size_t)64});
  HIP_TRY(hipSetDevice(handle->opts.device_id));
  HIP_TRY(hipHostMalloc((void**)&g->h_stage, g->h_stage_n * sizeof(double)));
  *out = g.release();
  return GMPNP_OK;
}

// Peer-mailbox transport, step 1: allocate this rank's mailbox and hand out its IPC handle.  The caller gathers the handles of
// all ranks (any channel: the Python driver uses torch.distributed.all_gather) and calls gmpnp_group_peer_connect.
int gmpnp_group_peer_begin(gmpnp_solver* handle, gmpnp_group** out, char ipc_handle[GMPNP_PEER_HANDLE_BYTES]) {
  static_assert(GMPNP_PEER_HANDLE_BYTES == sizeof(hipIpcMemHandle_t), "IPC handle size");
  if (!handle || !out || !ipc_handle) return fail(GMPNP_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (!handle->partitioned) return fail(GMPNP_ERR_INVALID, "group members must come from gmpnp_create_partition");
  if (handle->part_size > kPeerMax) return fail(GMPNP_ERR_INVALID, "peer transport: at most 8 ranks");
  if (handle->nb_rank.size() > (size_t)kPeerNbMax) return fail(GMPNP_ERR_INVALID, "peer transport: at most 8 neighbours per rank");
  std::unique_ptr<gmpnp_group> g(new gmpnp_group);
  g->dom.push_back(handle);
  g->peer = true;
  HIP_TRY(hipSetDevice(handle->opts.device_id));
  PeerArgs& a = g->pa;
  a.me = handle->part_rank; a.size = handle->part_size; a.seq = 0;
  const size_t n = (size_t)handle->ncoarse;
  a.red_cap = (int)std::max<size_t>({n * n, 2 + 3 * n, (size_t)8});
  a.red_cap = (a.red_cap + 15) & ~15;
  a.wmax = handle->nf * handle->nf;
  // mailbox layout (every offset the same on every rank; only the last area's size differs): flags | table | all-reduce contributions
  // | flagged-word sums [kLLSlots][size][red_cap] | flagged-word ghost rows [kLLHaloNodes][kLLSlots][kLLRow] (16 bytes a double) |
  // ghost rows of the flag-based exchanges (2 parities, wmax doubles per node)
  g->ll_red_off = kPeerRedOff + (size_t)2 * a.size * a.red_cap * sizeof(double);
  g->ll_halo_off = g->ll_red_off + (size_t)kLLSlots * a.size * a.red_cap * 16;
  a.halo_off = g->ll_halo_off + (size_t)kLLHaloNodes * kLLSlots * kLLRow * 16;
  a.n_nb = (int)handle->nb_rank.size();
  for (int j = 0; j < a.n_nb; ++j) a.nb_rank[j] = handle->nb_rank[j];
  for (int j = 0; j <= a.n_nb; ++j) { a.send_ptr[j] = handle->send_ptr[j]; a.recv_ptr[j] = handle->recv_ptr[j]; }
  g->box_bytes = a.halo_off + (size_t)2 * std::max(1, handle->recv_ptr.back()) * a.wmax * sizeof(double);
  // uncached: a peer's stores (and this rank's polls of them) must not meet a stale line in this GPU's L2
  HIP_TRY(hipExtMallocWithFlags((void**)&g->box, g->box_bytes, hipDeviceMallocUncached));
  HIP_TRY(hipMemset(g->box, 0, g->box_bytes));
  // where each neighbour's rows start in THIS rank's ghost-row area: the neighbour reads its entry after mapping the mailbox
  std::vector<int32_t> table(kPeerMax, -1);
  for (int j = 0; j < a.n_nb; ++j) table[a.nb_rank[j]] = a.recv_ptr[j];
  HIP_TRY(hipMemcpy(g->box + kPeerTableOff, table.data(), kPeerMax * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMalloc((void**)&g->peer_counter, sizeof(unsigned)));
  HIP_TRY(hipMemset(g->peer_counter, 0, sizeof(unsigned)));
  if (handle->fused_half && handle->nf == 9) {
    // the exchange may ride in front of the next launch's coarse workgroups where that launch is STILL resident at once
    int occ_a = 0, occ_b = 0, cus = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_a, k_half_a_x<9>, kKrylovThreads, 0));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, k_half_b_x<9>, kKrylovThreads, 0));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, handle->opts.device_id));
    const int nsn = handle->send_ptr.empty() ? 0 : handle->send_ptr.back();
    const int nx = std::max(xch_workgroups(2 + 3 * (int)n, nsn, 3, 9), xch_workgroups(4 + (int)n, nsn, 2, 9));
    g->prologue_ok = handle->t.own_ntiles + handle->t.nagg + nx <= std::min(occ_a, occ_b) * cus && handle->recv_ptr.back() <= kLLHaloNodes;
  }
  HIP_TRY(hipHostMalloc((void**)&g->h_peer_err, sizeof(int32_t)));
  *g->h_peer_err = 0;
  a.err = g->h_peer_err;
  int clock_khz = 0;   // wall_clock64 rate of THIS device (100 MHz on gfx950, not assumed)
  if (hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeWallClockRate, handle->opts.device_id) != hipSuccess || clock_khz <= 0) clock_khz = 100000;
  a.budget = 5ull * 1000ull * (unsigned long long)clock_khz;
  hipIpcMemHandle_t h;
  HIP_TRY(hipIpcGetMemHandle(&h, g->box));
  std::memcpy(ipc_handle, &h, sizeof h);
  *out = g.release();
  return GMPNP_OK;
}

// Step 2: map the other ranks' mailboxes (all_handles: [size][GMPNP_PEER_HANDLE_BYTES], rank order).  Every rank must have
// returned from gmpnp_group_peer_begin before any rank calls this (the gather of the handles is that point).
int gmpnp_group_peer_connect(gmpnp_group* g, const char* all_handles) {
  if (!g || !all_handles || !g->peer) return fail(GMPNP_ERR_INVALID, "bad arguments");
  if (g->peer_connected) return GMPNP_OK;
  gmpnp_solver* s = g->dom[0];
  HIP_TRY(hipSetDevice(s->opts.device_id));
  PeerArgs& a = g->pa;
  for (int q = 0; q < a.size; ++q) {
    if (q == a.me) { a.box[q] = g->box; continue; }
    hipIpcMemHandle_t h;
    std::memcpy(&h, all_handles + (size_t)q * GMPNP_PEER_HANDLE_BYTES, sizeof h);
    HIP_TRY(hipIpcOpenMemHandle(&g->peer_map[q], h, hipIpcMemLazyEnablePeerAccess));
    a.box[q] = (unsigned char*)g->peer_map[q];
  }
  for (int j = 0; j < a.n_nb; ++j) {
    int32_t off = -1;
    HIP_TRY(hipMemcpy(&off, a.box[a.nb_rank[j]] + kPeerTableOff + (size_t)a.me * sizeof(int32_t), sizeof off, hipMemcpyDeviceToHost));
    if (off < 0) return fail(GMPNP_ERR_INVALID, "peer transport: a neighbour's plan has no segment for this rank");
    a.peer_recv_ptr[j] = off;
  }
  g->peer_connected = true;
  return GMPNP_OK;
}

void gmpnp_group_destroy(gmpnp_group* g) {
  if (!g) return;
  if (g->peer) {   // (the caller has made sure that no rank is still inside an exchange: a barrier of its own)
    if (!g->dom.empty()) { (void)hipSetDevice(g->dom[0]->opts.device_id); (void)hipStreamSynchronize(g->dom[0]->stream); }
    for (int q = 0; q < kPeerMax; ++q) if (g->peer_map[q]) (void)hipIpcCloseMemHandle(g->peer_map[q]);
    if (g->box) (void)hipFree(g->box);
    if (g->peer_counter) (void)hipFree(g->peer_counter);
    if (g->h_peer_err) (void)hipHostFree(g->h_peer_err);
  }
  if (g->h_stage) (void)hipHostFree(g->h_stage);
  if (!g->dom.empty()) { (void)hipSetDevice(g->dom[0]->opts.device_id); (void)hipStreamSynchronize(g->dom[0]->stream); }
  for (size_t d = 1; d < g->own_stream.size(); ++d) if (g->own_stream[d]) g->dom[d]->stream = g->own_stream[d];
  delete g;
}

int gmpnp_group_newton_solve(gmpnp_group* g, const gmpnp_newton_options_t* o, gmpnp_newton_stats_t* stats) {
  if (!g || !o) return fail(GMPNP_ERR_INVALID, "NULL argument");
  if (o->maximum_iterations < 0 || o->krylov_maximum_iterations < 1) return fail(GMPNP_ERR_INVALID, "bad iteration limits");
  if (o->linear_solver != GMPNP_LINEAR_BICGSTAB_TWOLEVEL && o->linear_solver != GMPNP_LINEAR_BICGSTAB_JACOBI)
    return fail(GMPNP_ERR_INVALID, "the partitioned solve uses BiCGStab (two-level or Jacobi)");
  gmpnp_newton_stats_t local{};
  gmpnp_newton_stats_t& st = stats ? *stats : local;
  st = gmpnp_newton_stats_t{};
  HIP_TRY(hipSetDevice(g->dom[0]->opts.device_id));
  return group_newton<3, 9>(g, *o, st);
}

// One pass of each collective of the partitioned solve over the group's OWN transport (peer mailboxes, RCCL, host-staged or the
// in-process copies), with contents every rank can check by itself: a 5-double all-reduce of (rank + 1)(i + 1), and a ghost-row
// message per neighbour whose k-th value is sender * 1e6 + k.  Collective: every rank of the group calls it; *max_error = largest
// deviation seen by THIS process.  What the bench runs before it trusts a transport between physical GPUs with a timed solve.
int gmpnp_group_selftest(gmpnp_group* g, double* max_error) {
  if (!g || !max_error) return fail(GMPNP_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(g->dom[0]->opts.device_id));
  const int size = g->dom[0]->part_size;
  for (gmpnp_solver* s : g->dom) {
    double red[5];
    for (int i = 0; i < 5; ++i) red[i] = (double)(s->part_rank + 1) * (i + 1);
    HIP_TRY(hipMemcpyAsync(s->red_norm.p, red, sizeof red, hipMemcpyHostToDevice, s->stream));
    const int nsn = s->send_ptr.empty() ? 0 : s->send_ptr.back();
    std::vector<double> h((size_t)std::max(nsn, 1));
    for (size_t j = 0; j < s->nb_rank.size(); ++j)
      for (int k = s->send_ptr[j]; k < s->send_ptr[j + 1]; ++k) h[k] = 1e6 * s->part_rank + (k - s->send_ptr[j]);
    if (nsn) HIP_TRY(hipMemcpyAsync(s->sendbuf.p, h.data(), (size_t)nsn * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));   // (h leaves scope)
  }
  int rc = group_reduce_transfer(g, [](gmpnp_solver* s) { return s->red_norm.p; }, 5, 1); if (rc) return rc;
  double err = 0.0;
  for (gmpnp_solver* s : g->dom) {
    // in-process groups run every handle on dom[0]'s stream
    double red[5];
    HIP_TRY(hipMemcpyAsync(red, s->red_norm.p, sizeof red, hipMemcpyDeviceToHost, g->dom[0]->stream));
    const int nrn = s->recv_ptr.empty() ? 0 : s->recv_ptr.back();
    std::vector<double> h((size_t)std::max(nrn, 1), 0.0);
    if (nrn) HIP_TRY(hipMemcpyAsync(h.data(), s->recvbuf.p, (size_t)nrn * sizeof(double), hipMemcpyDeviceToHost, g->dom[0]->stream));
    HIP_TRY(hipStreamSynchronize(g->dom[0]->stream));
    for (int i = 0; i < 5; ++i) err = std::max(err, std::fabs(red[i] - 0.5 * size * (size + 1) * (i + 1)));
    for (size_t j = 0; j < s->nb_rank.size(); ++j)
      for (int k = s->recv_ptr[j]; k < s->recv_ptr[j + 1]; ++k) err = std::max(err, std::fabs(h[k] - (1e6 * s->nb_rank[j] + (k - s->recv_ptr[j]))));
  }
  if (g->peer && *g->h_peer_err) return fail(GMPNP_ERR_HIP, "peer transport: a rank's flag did not arrive within 5 s");
  if (g->peer && g->prologue_ok && g->exchange_form != 1 && g->dom[0]->fused_half) {
    // ... and the flagged-word areas the exchange-prologue launches of a solve use (k_xch_selftest), over the same mapping
    gmpnp_solver* s = g->dom[0];
    XchArgs x = make_xch_args(g, 0, 0, 0, VecList{}, 0);
    Ctx cc = xch_ctx(g, x);
    HIP_TRY(hipMemsetAsync(s->status.p, 0, sizeof(int32_t), s->stream));
    hipLaunchKernelGGL(k_xch_selftest, dim3(1), dim3(kKrylovThreads), 0, s->stream, cc, x, g->pa, s->red_norm.p);
    HIP_TRY(hipGetLastError());
    double xerr = 0.0; int32_t st = 0;
    HIP_TRY(hipMemcpyAsync(&xerr, s->red_norm.p, sizeof xerr, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(&st, s->status.p, sizeof st, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemsetAsync(s->status.p, 0, sizeof(int32_t), s->stream));
    if (st & 8) return fail(GMPNP_ERR_HIP, "peer transport: a flagged word of another rank did not arrive within 12 s");
    err = std::max(err, xerr);
  }
  *max_error = err;
  return GMPNP_OK;
}

int gmpnp_group_set_exchange_form(gmpnp_group* g, int32_t form) {
  if (!g || (form != 0 && form != 1)) return fail(GMPNP_ERR_INVALID, "bad arguments");
  g->exchange_form = form;
  return GMPNP_OK;
}
int32_t gmpnp_group_exchange_form(const gmpnp_group* g) {
  if (!g) return -1;
  return (g->peer && g->prologue_ok && g->exchange_form != 1 && g->dom[0]->fused_half) ? 2 : (g->peer ? 1 : 0);
}

int gmpnp_group_assign_previous(gmpnp_group* g) {
  if (!g) return fail(GMPNP_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(g->dom[0]->opts.device_id));
  for (gmpnp_solver* s : g->dom) HIP_TRY(hipMemcpyAsync(s->un.p, s->u.p, s->ndof * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
  return GMPNP_OK;
}

}  // extern "C"
