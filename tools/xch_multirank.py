"""Peer transport between PROCESSES sharing the box's one GPU, L_50_R_5, first Newton solve from the zero state: exchange form 0
(flagged words in front of the next launch) against form 1 (separate exchange launches) at 2 and 4 ranks — same Newton counts,
states equal to 1e-9, wall-clock of the solve.  (The ranks time-slice one card: the times say what the protocol costs there, not
how N GPUs scale.)   python tools/xch_multirank.py [L_nm R_nm]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

MUMPS_09 = {"nonlinear_solver": "newton", "newton_solver": {"linear_solver": "mumps", "maximum_iterations": 50, "relative_tolerance": 1e-4,
                                                             "absolute_tolerance": 1e-4, "relaxation_parameter": 0.9}}


def worker(rank, world, port, out, form, L, R):
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gmpnp_amd import backend, dist
        from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
        from gmpnp_amd.params import pore_parameters, utilities_dir
        from gmpnp_amd.problem import pore_problem
        pp = pore_parameters(concentration_elec=0.5, L=L * 1e-9, R=R * 1e-9)
        mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
        prob, _ = pore_problem(pp, mesh)
        nv = mesh.num_vertices
        with dist.PartitionedSolver(prob, world, rank=rank, transport="peer", exchange_form=form) as ps:
            f = ps.exchange_form()
            assert ps.selftest() == 0.0
            walls = []
            for rep in range(3):
                ps.set_state(np.zeros(nv * 9), np.tile(np.r_[np.ones(8), 0.0], nv))
                tdist.barrier()
                t0 = time.perf_counter()
                st = ps.newton_solve(backend.newton_options(MUMPS_09))
                walls.append(time.perf_counter() - t0)
            u = ps.get_state()
        if rank == 0:
            np.savez(out, u=u, its=st["iterations"], kits=np.array(st["krylov_per_iteration"]), wall=min(walls), form=f)
    finally:
        tdist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    L, R = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (50.0, 5.0)
    tmp = tempfile.mkdtemp()
    res = {}
    for world in [int(w) for w in os.environ.get("XCH_WORLDS", "2,4").split(",")]:   # (a GPU box admits 6 processes on its card)
        for form in (1, 0):
            out = os.path.join(tmp, "w%d_f%d.npz" % (world, form))
            mp.spawn(worker, args=(world, 29800 + 7 * world + form, out, form, L, R), nprocs=world, join=True)
            d = np.load(out)
            res[(world, form)] = d
            print("ranks %d  form asked %d -> runs %d | Newton %d, BiCGStab %d, solve %.1f ms, %.1f us per BiCGStab iteration"
                  % (world, form, int(d["form"]), int(d["its"]), int(d["kits"].sum()), 1e3 * float(d["wall"]), 1e6 * float(d["wall"]) / d["kits"].sum()), flush=True)
        a, b = res[(world, 0)]["u"], res[(world, 1)]["u"]
        print("ranks %d  state form 0 vs form 1: %.2e" % (world, np.linalg.norm(a - b) / np.linalg.norm(b)), flush=True)
