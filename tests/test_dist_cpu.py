"""World-size-2 gloo test of what the partitioned solve needs from Python (gmpnp_amd/dist.py): the partition, the plan handed
to gmpnp_create_partition and the host-transport callbacks the library calls back into.  The Newton / BiCGStab iteration that
drives them here is a NumPy test double (tests/partition_double.py) on the CPU oracle's local operations; the library's own
loops (csrc/gmpnp_group.h) run in the GPU tests — two and four PROCESSES over the same callbacks among them
(tests/test_gpu_parity.py::test_library_partitioned_solve_two_processes_on_one_card)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_problem():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import box_pore_problem
    return box_pore_problem(nx=4, nz=16)[2]  # 425 vertices


def test_partition_and_halo_plan():
    from gmpnp_amd import dist
    prob = _small_problem()
    nv = prob.coords.shape[0]
    for P in (2, 4):
        owner = dist.slab_owner(prob.coords, prob.cells, P)
        counts = np.bincount(owner, minlength=P)
        assert counts.sum() == nv and counts.max() - counts.min() <= 1
        doms = [dist.build_local_domain(prob, owner, r, P) for r in range(P)]
        assert sum(d.n_owned for d in doms) == nv
        for d in doms:
            # every cell touching an owned vertex is local, ghost rows are Dirichlet rows
            gh = np.arange(d.n_owned, d.n_owned + len(d.ghosts))
            assert np.isin((gh[:, None] * 9 + np.arange(9)).ravel(), d.problem.bc_dofs).all()
            for q, idx in d.recv.items():
                # what q sends me is exactly my ghosts owned by q, in the same order
                sent_global = doms[q].owned[doms[q].send[d.rank]]
                assert np.array_equal(sent_global, d.ghosts[idx - d.n_owned])
            assert len(d.ghosts) <= 4 * 25  # two interfaces, each at most two (partial) 5x5 vertex layers
        # slabs only talk to their neighbours along the axis
        assert all(abs(q - d.rank) == 1 for d in doms for q in d.recv)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import partition_double as pd
        from gmpnp_amd import dist
        prob = _small_problem()
        nv, nf = prob.coords.shape[0], prob.nf
        dom, perm, part = dist.partition_plan(prob, world, rank)          # what gmpnp_create_partition gets
        n_local = dom.n_owned + len(dom.ghosts)
        comm = pd.PlanComm(part, n_local, nf, dist.host_transport_callbacks())   # the PRODUCT's callbacks, the library's buffer layout
        ops = pd.OracleLocalOps(dom.problem)
        # the whole first Newton solve of time step 0 (zero initial guess, as the reference)
        un = np.tile(np.r_[np.ones(8), 0.0], nv)
        u0, un0 = dist.scatter_local(dom, np.zeros(nv * nf)), dist.scatter_local(dom, un)
        u, st = pd.newton_solve(ops, comm, dom.n_owned * nf, u0, un0, relaxation_parameter=0.9, krylov_rtol=1e-11, krylov_maxit=3000)
        ug = np.zeros((nv, nf))
        ug[dom.owned] = u.reshape(-1, nf)[:dom.n_owned]
        ug = comm.allreduce_sum(ug.ravel())                                # disjoint owned pieces
        if rank == 0:
            np.savez(os.path.join(out_dir, "dist.npz"), u=ug, its=st["iterations"], res=np.array(st["residuals"]),
                     kits=np.array(st["krylov_per_iteration"]))
    finally:
        tdist.destroy_process_group()


def test_partitioned_newton_matches_serial_oracle(tmp_path):
    import socket
    import torch.multiprocessing as mp
    import gmpnp_oracle as O
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "dist.npz"))
    prob = _small_problem()
    nv = prob.coords.shape[0]
    un = np.tile(np.r_[np.ones(8), 0.0], nv)
    u_ref, st = O.newton_solve(prob, np.zeros(nv * 9), un, relaxation_parameter=0.9)
    assert int(got["its"]) == st.iterations and st.iterations >= 5
    # serial reference = sparse LU; the partitioned Krylov runs to 1e-11 (the u = 0 Jacobian of step 0 is ill conditioned)
    print("max rel residual-history diff", np.max(np.abs(got["res"] / np.array(st.residuals) - 1)),
          "state diff", np.linalg.norm(got["u"] - u_ref) / np.linalg.norm(u_ref), "krylov", got["kits"])
    assert np.allclose(got["res"], st.residuals, rtol=1e-5)
    assert np.linalg.norm(got["u"] - u_ref) / np.linalg.norm(u_ref) < 1e-8
    assert got["kits"].max() < 3000


@pytest.mark.parametrize("nparts", [1, 2, 4, 8])
def test_partition_plan_is_consistent(nparts, pore10):
    """The plan gmpnp_create_partition gets (dist.partition_plan): ownership ranges and coarse slabs share their
    boundaries, the local vertex order runs through the slabs in ascending order with the owned vertices contiguous, and
    what rank p sends to q is, entry by entry, what q expects from p."""
    from gmpnp_amd import dist
    pp, mesh, prob, _ = pore10
    plans = [dist.partition_plan(prob, nparts, r) for r in range(nparts)]
    nv = mesh.num_vertices
    seen = np.zeros(nv, dtype=int)
    for r, (dom, perm, part) in enumerate(plans):
        lverts = np.concatenate([dom.owned, dom.ghosts])
        seen[dom.owned] += 1
        ag, own = part["vertex_aggregate"][perm], part["vertex_owned"][perm]
        assert (np.diff(ag) >= 0).all()
        first, last = np.nonzero(own)[0][[0, -1]]
        assert own[first:last + 1].all() and own.sum() == dom.n_owned
        assert set(ag[own == 1]).isdisjoint(set(ag[own == 0]))                      # a slab is all owned or all ghost
        assert part["n_global_aggregates"] % nparts == 0
        for j, q in enumerate(part["neighbour_rank"]):
            qd, _, qp = plans[q]
            jj = list(qp["neighbour_rank"]).index(r)
            mine = lverts[part["send_vertices"][part["send_ptr"][j]:part["send_ptr"][j + 1]]]
            ql = np.concatenate([qd.owned, qd.ghosts])
            theirs = ql[qp["recv_vertices"][qp["recv_ptr"][jj]:qp["recv_ptr"][jj + 1]]]
            assert len(mine) > 0 and np.array_equal(mine, theirs)                   # same global vertices, same order
    assert (seen == 1).all()
