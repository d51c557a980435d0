"""Mesh partitioning for the multi-GPU solve (BASELINE configs[3]): one process per GPU.

The reference is a serial script (no MPI call site), so there is no behaviour to match except the serial result itself: the
partitioned solve must give the serial Newton iterates.

Decomposition.  Vertices are ordered by `backend.slab_permutation` (slabs along the pore axis) and cut into P contiguous
ranges: rank p OWNS its range.  Its local mesh is every cell that touches an owned vertex; the other vertices of those cells
are GHOSTS (owned by a neighbouring slab).  Cut cells are assembled redundantly on both sides, so the rows of owned vertices
are complete without any matrix communication; ghost rows are replaced by identity rows (they are never used).

What this module is:
* ``partition_plan`` / ``build_local_domain`` — the partition, the halo plan and the global coarse slabs handed to
  ``gmpnp_create_partition``;
* ``PartitionedSolver`` — the product path: Newton, BiCGStab, ghost exchanges and all-reduces run INSIDE libgmpnp.so
  (``gmpnp_group_newton_solve``; RCCL, in-process, or host-staged transport); Python scatters / gathers states and the
  per-step boundary values;
* ``host_transport_callbacks`` — the two collectives of the library's host-staged transport on ``torch.distributed``.
(Round 1's host-driven Python BiCGStab / Newton is gone from the product; tests/partition_double.py keeps a NumPy double of the
partitioned iteration that drives ``partition_plan`` and these callbacks at world size 2 on ``gloo`` without a GPU.)
"""
from __future__ import annotations

import copy
from dataclasses import dataclass

import numpy as np

from .problem import Problem


# ---------------------------------------------------------------------------------------------
# partition
# ---------------------------------------------------------------------------------------------
def slab_owner(coords: np.ndarray, cells: np.ndarray, nparts: int) -> np.ndarray:
    """owner[vertex] for `nparts` equal-count slabs of the slab order (the order the solver uses internally)."""
    from .backend import slab_permutation
    perm = slab_permutation(coords, cells, window=0)  # pure slab order: sharp partition interfaces
    owner = np.empty(coords.shape[0], dtype=np.int32)
    nv = coords.shape[0]
    for p in range(nparts):
        owner[perm[(nv * p) // nparts:(nv * (p + 1)) // nparts]] = p
    return owner


@dataclass
class LocalDomain:
    rank: int
    nparts: int
    owned: np.ndarray  # global (file) vertex ids, ascending
    ghosts: np.ndarray  # global vertex ids, ascending
    ghost_owner: np.ndarray  # rank owning each ghost
    problem: Problem  # local problem: vertices = [owned..., ghosts...]
    n_owned: int
    send: dict  # neighbour rank -> local indices (into owned) to send, in the receiver's ghost order
    recv: dict  # neighbour rank -> local indices (n_owned + k) that receive

    @property
    def nf(self):
        return self.problem.nf

    def owned_dofs(self):
        return slice(0, self.n_owned * self.nf)


def build_local_domain(prob: Problem, owner: np.ndarray, rank: int, nparts: int) -> LocalDomain:
    """Local problem of `rank`.  Deterministic and purely local (every rank can build any rank's domain), so the
    send lists need no negotiation: rank q's ghosts owned by p, in ascending global id, are what p sends to q."""
    nf = prob.nf
    cells = prob.cells
    touch = (owner[cells] == rank).any(axis=1)
    lcells = cells[touch]
    verts = np.unique(lcells)
    owned = verts[owner[verts] == rank]
    ghosts = verts[owner[verts] != rank]
    all_owned = np.nonzero(owner == rank)[0]
    assert np.array_equal(owned, all_owned), "every owned vertex must belong to a local cell"
    lverts = np.concatenate([owned, ghosts])
    g2l = -np.ones(prob.coords.shape[0], dtype=np.int64)
    g2l[lverts] = np.arange(len(lverts))

    def facets_local(fv):
        if len(fv) == 0:
            return np.zeros((0, 3), dtype=np.int32)
        keep = (owner[fv] == rank).any(axis=1)
        return g2l[fv[keep]].astype(np.int32)

    # Dirichlet: the global conditions restricted to local vertices + identity rows on every ghost dof
    gv = prob.bc_dofs // nf
    inloc = g2l[gv] >= 0
    ldofs = g2l[gv[inloc]] * nf + prob.bc_dofs[inloc] % nf
    lvals = prob.bc_vals[inloc]
    gh_dofs = (np.arange(len(owned), len(lverts))[:, None] * nf + np.arange(nf)[None, :]).ravel()
    table = dict(zip(ldofs.tolist(), lvals.tolist()))
    for d in gh_dofs.tolist():
        table[d] = 0.0  # value irrelevant: ghost rows are never used
    bd = np.array(sorted(table), dtype=np.int64)
    bvl = np.array([table[d] for d in bd])
    pv = prob.point_vertices
    pv_local = g2l[pv[owner[pv] == rank]].astype(np.int32) if len(pv) else np.zeros(0, dtype=np.int32)
    local = Problem(coords=prob.coords[lverts], cells=g2l[lcells].astype(np.int32), model=prob.model, quad=prob.quad,
                    wall_facets=facets_local(prob.wall_facets), exit_facets=facets_local(prob.exit_facets),
                    point_vertices=pv_local, bc_dofs=bd, bc_vals=bvl)
    # halo plan
    send, recv = {}, {}
    gowner = owner[ghosts]
    for q in np.unique(gowner):
        recv[int(q)] = len(owned) + np.nonzero(gowner == q)[0]
    for q in range(nparts):
        if q == rank:
            continue
        tq = (owner[cells] == q).any(axis=1)
        vq = np.unique(cells[tq])
        mine = vq[owner[vq] == rank]  # ascending global id == q's ghost order restricted to my vertices
        if len(mine):
            send[q] = g2l[mine]
    return LocalDomain(rank=rank, nparts=nparts, owned=owned, ghosts=ghosts, ghost_owner=gowner, problem=local,
                       n_owned=len(owned), send=send, recv=recv)


# ---------------------------------------------------------------------------------------------
# the partitioned solve INSIDE the library (gmpnp_create_partition / gmpnp_group_*): this module only partitions and plans
# ---------------------------------------------------------------------------------------------
def default_global_aggregates(nparts: int) -> int:
    """Coarse slabs over the whole mesh: 8 (the single-GPU default) when the ranks divide it, else one slab per rank
    rounded up to a multiple of the rank count; at most 15 (the 9-field coarse operator must fit the LDS-resident inverse)."""
    if 8 % nparts == 0:
        return 8
    n = nparts * max(1, 8 // nparts)
    if n > 15:
        raise ValueError("no coarse-slab count <= 15 is a multiple of %d ranks" % nparts)
    return n


def partition_plan(prob: Problem, nparts: int, rank: int, n_global_aggregates: int = None):
    """Everything rank `rank` needs for ``gmpnp_create_partition``: (LocalDomain, local perm, partition dict).

    Global slab order (``backend.slab_permutation``: vertices sorted along the pore axis) is cut into `nparts` contiguous
    ownership ranges and into `n_global_aggregates` coarse slabs with the SAME integer boundaries, so a slab never
    straddles two ranks.  The local vertex order handed to the library is the global slab order restricted to the local
    vertices: ghosts of the lower neighbour, owned vertices, ghosts of the upper neighbour."""
    from .backend import slab_permutation
    nag = n_global_aggregates or default_global_aggregates(nparts)
    if nag % nparts:
        raise ValueError("n_global_aggregates must be a multiple of the number of ranks")
    nv = prob.coords.shape[0]
    gperm = slab_permutation(prob.coords, prob.cells, window=0)
    pos = np.empty(nv, dtype=np.int64)
    pos[gperm] = np.arange(nv)
    bounds = (nv * np.arange(nag + 1, dtype=np.int64)) // nag
    agg_of = np.searchsorted(bounds[1:], pos, side="right").astype(np.int32)       # slab of every global vertex
    owner = (agg_of // (nag // nparts)).astype(np.int32)
    dom = build_local_domain(prob, owner, rank, nparts)
    lverts = np.concatenate([dom.owned, dom.ghosts])
    perm_local = np.argsort(pos[lverts], kind="stable").astype(np.int32)
    nbrs = sorted(set(dom.send) | set(dom.recv))
    send_ptr, recv_ptr, send_v, recv_v = [0], [0], [], []
    for q in nbrs:
        send_v.extend(np.asarray(dom.send.get(q, []), dtype=np.int64).tolist())
        recv_v.extend(np.asarray(dom.recv.get(q, []), dtype=np.int64).tolist())
        send_ptr.append(len(send_v))
        recv_ptr.append(len(recv_v))
    owned_flag = np.zeros(len(lverts), dtype=np.uint8)
    owned_flag[:dom.n_owned] = 1
    part = {"rank": rank, "size": nparts, "n_global_aggregates": nag, "vertex_aggregate": agg_of[lverts], "vertex_owned": owned_flag,
            "neighbour_rank": np.array(nbrs, dtype=np.int32), "send_ptr": np.array(send_ptr, dtype=np.int32),
            "send_vertices": np.array(send_v, dtype=np.int32), "recv_ptr": np.array(recv_ptr, dtype=np.int32),
            "recv_vertices": np.array(recv_v, dtype=np.int32)}
    return dom, perm_local, part


class PartitionedSolver:
    """One mesh-partitioned problem solved by libgmpnp.so across `nparts` ranks (SURVEY section 8e / BASELINE configs[3]).

    ``PartitionedSolver(prob, nparts)``                every rank in THIS process on one GPU (rehearsal: the exchanges are device
                                                       copies between the handles) — what a single-GPU box can run;
    ``PartitionedSolver(prob, nparts, rank=r, ...)``   one rank per process / GPU; the RCCL communicator is created inside the
                                                       library from an id that rank 0 makes and ``torch.distributed`` broadcasts.
    The Krylov and Newton loops run in the library; Python only scatters / gathers states and per-step boundary values."""

    def __init__(self, prob: Problem, nparts: int, rank: int = None, device_id: int = 0, n_global_aggregates: int = None,
                 use_torch_dist: bool = True, transport: str = "rccl", exchange_form: int = 0, **device_kwargs):
        """``transport`` (one rank per process only): "peer" — peer mailboxes: every collective is one kernel launch per rank
        that stores into the other ranks' IPC-mapped mailboxes (xGMI between GPUs; also works for ranks sharing a card);
        "rccl" — collectives inside the library over RCCL; "host" — the library stages every collective through pinned host
        memory and calls back into ``torch.distributed`` (any backend, e.g. gloo).  ``exchange_form`` (peer transport): 0 = the
        exchange of a half-iteration rides inside the next launch where that launch is resident (2 launches per BiCGStab iteration),
        1 = separate exchange launches (4); every rank must pass the same value (gmpnp_group_set_exchange_form)."""
        from ctypes import byref, c_void_p, create_string_buffer
        if rank is not None:
            # one rank per process: torch.distributed carries the set-up (mailbox handles, communicator id, host-staged collectives).
            # PyTorch first, THEN libgmpnp.so: the library then binds to the HIP runtime PyTorch ships instead of bringing the
            # system's into the same process (README: two HIP runtimes on one GPU do not mix)
            import torch  # noqa: F401
        from . import backend
        self.backend = backend
        self.lib = backend.load_library()
        self.nparts, self.rank = nparts, rank
        self.nv_global, self.nf = prob.coords.shape[0], prob.nf
        self.ranks = list(range(nparts)) if rank is None else [rank]
        self.doms, self.devs = [], []
        for r in self.ranks:
            dom, perm, part = partition_plan(prob, nparts, r, n_global_aggregates)
            self.doms.append(dom)
            self.devs.append(backend.DeviceSolver(dom.problem, device_id=device_id, perm=perm, partition=part, **device_kwargs))
        self._comm = c_void_p()
        self._group = c_void_p()
        self.transport = transport if rank is not None else "in-process"
        if rank is not None and transport == "peer":
            import torch
            import torch.distributed as tdist
            if not (tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() == nparts):
                raise RuntimeError("the peer transport needs an initialised torch.distributed group of %d ranks (to gather the mailbox handles)" % nparts)
            # Every step is agreed on by ALL ranks before anyone goes on: a rank that cannot allocate or map a mailbox (no peer
            # access to a GPU, IPC refused) must not leave the others waiting in a collective it never joins.
            mine = create_string_buffer(backend.PEER_HANDLE_BYTES)
            err = None
            try:
                self._check(self.lib.gmpnp_group_peer_begin(self.devs[0]._h, byref(self._group), mine))
            except backend.GmpnpError as e:
                err = e
            dev = "cuda" if tdist.get_backend() == "nccl" else "cpu"
            t = torch.tensor(list(mine.raw) + [0 if err is None else 1], dtype=torch.uint8, device=dev)
            parts = [torch.empty_like(t) for _ in range(nparts)]
            tdist.all_gather(parts, t)      # also the point after which every rank's mailbox exists
            rows = [x.cpu().tolist() for x in parts]
            if err is None and not any(r[-1] for r in rows):
                allh = create_string_buffer(backend.PEER_HANDLE_BYTES * nparts)
                allh.raw = b"".join(bytes(r[:-1]) for r in rows)
                try:
                    self._check(self.lib.gmpnp_group_peer_connect(self._group, allh))
                except backend.GmpnpError as e:
                    err = e
            elif err is None:
                err = RuntimeError("rank(s) %s could not set up a mailbox" % [q for q, r in enumerate(rows) if r[-1]])
            flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=dev)
            tdist.all_reduce(flag, op=tdist.ReduceOp.MAX)   # (also: every mailbox is mapped everywhere from here on)
            if int(flag[0]):
                self.transport = "peer (failed)"
                self.close()
                raise RuntimeError("peer-mailbox transport not available: %s" % (err if err is not None else "another rank could not map a mailbox"))
            # All ranks must run the SAME form (form 2 leaves the exchange behind the last launch of a solve out, so the sequence
            # numbers of the two forms drift apart): a rank whose launch would not be resident with the exchange workgroups in
            # front takes everybody back to separate launches.
            self._check(self.lib.gmpnp_group_set_exchange_form(self._group, int(exchange_form)))
            form = torch.tensor([int(self.lib.gmpnp_group_exchange_form(self._group))], dtype=torch.int32, device=dev)
            tdist.all_reduce(form, op=tdist.ReduceOp.MIN)
            if int(form[0]) != 2:
                self._check(self.lib.gmpnp_group_set_exchange_form(self._group, 1))
            return
        if rank is not None and transport == "host":
            self._make_host_transport(rank, nparts)
            self._check(self.lib.gmpnp_group_create_hosted(self.devs[0]._h, byref(self._host_transport), byref(self._group)))
            return
        if rank is not None:
            idbuf = create_string_buffer(backend.COMM_ID_BYTES)
            if nparts > 1 or use_torch_dist:
                import torch
                import torch.distributed as tdist
                if tdist.is_available() and tdist.is_initialized():
                    if rank == 0:
                        self._check(self.lib.gmpnp_comm_unique_id(idbuf))
                    dev = "cuda" if tdist.get_backend() == "nccl" else "cpu"
                    t = torch.tensor(list(idbuf.raw), dtype=torch.uint8, device=dev)
                    tdist.broadcast(t, src=0)
                    idbuf = create_string_buffer(backend.COMM_ID_BYTES)
                    idbuf.raw = bytes(t.cpu().tolist())
                elif nparts == 1:
                    self._check(self.lib.gmpnp_comm_unique_id(idbuf))
                else:
                    raise RuntimeError("torch.distributed is not initialised: the communicator id cannot reach the other ranks")
            else:
                self._check(self.lib.gmpnp_comm_unique_id(idbuf))
            self._check(self.lib.gmpnp_comm_create(idbuf, rank, nparts, device_id, byref(self._comm)))
        handles = (c_void_p * len(self.devs))(*[d._h for d in self.devs])
        self._check(self.lib.gmpnp_group_create(len(self.devs), handles, self._comm if rank is not None else None, byref(self._group)))

    def _check(self, code):
        if code != self.backend.OK:
            raise self.backend.GmpnpError(code, self.lib.gmpnp_last_error().decode())

    def _make_host_transport(self, rank, nparts):
        """The two callbacks of gmpnp_host_transport_t on torch.distributed (CPU tensors that alias the library's pinned
        staging buffers: no copies on this side)."""
        import torch
        import torch.distributed as tdist
        backend = self.backend
        if not (tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() == nparts):
            raise RuntimeError("the host transport needs an initialised torch.distributed group of %d ranks" % nparts)

        # host tensors need a CPU-capable backend: a gloo group next to an nccl default group
        grp = tdist.new_group(backend="gloo") if tdist.get_backend() == "nccl" else None
        self._host_group = grp

        def view(ptr, n):
            return np.ctypeslib.as_array(ptr, shape=(int(n),))

        ar, ex = host_transport_callbacks(grp)

        def allreduce(_user, buf, n):
            return ar(view(buf, n))

        def exchange(_user, n_nb, nb_rank, s_off, s_cnt, s_buf, r_off, r_cnt, r_buf):
            n_s = max([0] + [s_off[j] + s_cnt[j] for j in range(n_nb)])
            n_r = max([0] + [r_off[j] + r_cnt[j] for j in range(n_nb)])
            return ex([int(nb_rank[j]) for j in range(n_nb)], [int(s_off[j]) for j in range(n_nb)], [int(s_cnt[j]) for j in range(n_nb)],
                      view(s_buf, max(n_s, 1)), [int(r_off[j]) for j in range(n_nb)], [int(r_cnt[j]) for j in range(n_nb)], view(r_buf, max(n_r, 1)))

        self._cb = (backend.ALLREDUCE_FN(allreduce), backend.EXCHANGE_FN(exchange))   # keep the thunks alive
        t = backend.CHostTransport()
        t.rank, t.size, t.allreduce, t.exchange, t.user = rank, nparts, self._cb[0], self._cb[1], None
        self._host_transport = t

    # ---- state in GLOBAL (file) vertex order -------------------------------------------------------------------------
    def set_state(self, u_global=None, un_global=None):
        for dom, dev in zip(self.doms, self.devs):
            dev.set_state(None if u_global is None else scatter_local(dom, np.asarray(u_global)),
                          None if un_global is None else scatter_local(dom, np.asarray(un_global)))

    def set_dirichlet(self, dofs, vals):
        """The GLOBAL Dirichlet set (file-order dofs); every local handle gets its part + identity rows on its ghost dofs
        (their values never enter anything: ghost rows are masked out of the residual)."""
        nf = self.nf
        dofs, vals = np.asarray(dofs, dtype=np.int64), np.asarray(vals, dtype=np.float64)
        if not hasattr(self, "_g2l"):
            self._g2l, self._ghost_dofs = [], []
            for dom in self.doms:
                lverts = np.concatenate([dom.owned, dom.ghosts])
                g2l = -np.ones(self.nv_global, dtype=np.int64)
                g2l[lverts] = np.arange(len(lverts))
                self._g2l.append(g2l)
                self._ghost_dofs.append(np.arange(dom.n_owned * nf, len(lverts) * nf, dtype=np.int64))
        for g2l, gh, dom, dev in zip(self._g2l, self._ghost_dofs, self.doms, self.devs):
            lv = g2l[dofs // nf]
            keep = (lv >= 0) & (lv < dom.n_owned)
            dev.set_dirichlet(np.concatenate([lv[keep] * nf + dofs[keep] % nf, gh]), np.concatenate([vals[keep], np.zeros(len(gh))]))

    def owned_state(self, previous=False):
        """[(owned global vertex ids, (n_owned, nf) values)] of the local ranks."""
        out = []
        for dom, dev in zip(self.doms, self.devs):
            out.append((dom.owned, dev.get_state(previous).reshape(-1, self.nf)[:dom.n_owned]))
        return out

    def get_state(self):
        """Global state (file order) — complete in the in-process form; with one rank per process all-gathered through
        torch.distributed when it is initialised, else only this rank's rows are filled."""
        out = np.zeros((self.nv_global, self.nf))
        for ids, vals in self.owned_state():
            out[ids] = vals
        if self.rank is not None and self.nparts > 1:
            import torch
            import torch.distributed as tdist
            if tdist.is_available() and tdist.is_initialized():
                dev = "cuda" if tdist.get_backend() == "nccl" else "cpu"
                t = torch.from_numpy(out).to(dev)
                tdist.all_reduce(t)
                out = t.cpu().numpy()
        return out.ravel()

    def comm_selftest(self, n=4096):
        """Send-to-self + receive + all-reduce through the library's RCCL bindings; returns the largest error."""
        from ctypes import byref, c_double
        err = c_double()
        self._check(self.lib.gmpnp_comm_selftest(self._comm, n, byref(err)))
        return err.value

    def exchange_form(self):
        """2 = the peer exchange rides inside the next half-iteration's launch, 1 = separate launches, 0 = another transport."""
        return int(self.lib.gmpnp_group_exchange_form(self._group))

    def selftest(self):
        """One all-reduce and one ghost-row exchange with self-checking contents over this group's transport
        (gmpnp_group_selftest; collective); returns the largest deviation this process saw (0.0 expected)."""
        from ctypes import byref, c_double
        err = c_double(-1.0)
        self._check(self.lib.gmpnp_group_selftest(self._group, byref(err)))
        return err.value

    def newton_solve(self, options, error_on_nonconvergence=True):
        from ctypes import byref
        st = self.backend.CNewtonStats()
        code = self.lib.gmpnp_group_newton_solve(self._group, byref(options), byref(st))
        stats = self.backend.DeviceSolver.stats_dict(st)
        if code == self.backend.ERR_NOT_CONVERGED and not error_on_nonconvergence:
            return stats
        self._check(code)
        return stats

    def assign_previous(self):
        self._check(self.lib.gmpnp_group_assign_previous(self._group))

    def close(self):
        if getattr(self, "transport", None) == "peer" and getattr(self, "_group", None):
            try:   # no rank unmaps / frees a mailbox another rank may still store into
                import torch.distributed as tdist
                if tdist.is_available() and tdist.is_initialized():
                    tdist.barrier()
            except Exception:  # noqa: BLE001
                pass
        if getattr(self, "_group", None):
            self.lib.gmpnp_group_destroy(self._group)
            self._group = None
        for d in getattr(self, "devs", []):
            d.close()
        self.devs = []
        if getattr(self, "_comm", None) and self._comm.value:
            self.lib.gmpnp_comm_destroy(self._comm)
            self._comm = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def host_transport_callbacks(group=None):
    """The two collectives of ``gmpnp_host_transport_t`` on ``torch.distributed`` (any backend with CPU tensors, e.g. gloo), as
    functions of NumPy arrays that ALIAS the caller's buffers (the library's pinned staging memory: no copies on this side):
    ``allreduce(buf)`` sums in place over the ranks; ``exchange(nb_rank, s_off, s_cnt, s_buf, r_off, r_cnt, r_buf)`` sends
    ``s_buf[s_off[j] : s_off[j] + s_cnt[j]]`` to neighbour j and receives ``r_buf[r_off[j] : ...]`` from it.  Both return 0, or 1
    after printing the traceback (the library turns that into GMPNP_ERR_HIP)."""
    import torch
    import torch.distributed as tdist

    def allreduce(buf):
        try:
            tdist.all_reduce(torch.from_numpy(buf), group=group)
            return 0
        except Exception:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    def exchange(nb_rank, s_off, s_cnt, s_buf, r_off, r_cnt, r_buf):
        try:
            sv, rv = torch.from_numpy(s_buf), torch.from_numpy(r_buf)
            ops = []
            for j, q in enumerate(nb_rank):
                if s_cnt[j]:
                    ops.append(tdist.P2POp(tdist.isend, sv[s_off[j]:s_off[j] + s_cnt[j]], int(q), group=group))
                if r_cnt[j]:
                    ops.append(tdist.P2POp(tdist.irecv, rv[r_off[j]:r_off[j] + r_cnt[j]], int(q), group=group))
            for req in (tdist.batch_isend_irecv(ops) if ops else []):
                req.wait()
            return 0
        except Exception:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    return allreduce, exchange


def scatter_local(dom: LocalDomain, u_global):
    nf = dom.nf
    lverts = np.concatenate([dom.owned, dom.ghosts])
    return u_global.reshape(-1, nf)[lverts].ravel().copy()
