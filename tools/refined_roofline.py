"""Kernel times on uniformly refined copies of L_50_R_5 (GPU box): where does the solver leave the cache-resident,
launch-bound regime?  Prints per level: sizes, back-to-back kernel times and algorithmic GB/s of the SpMV-centred kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from gmpnp_amd import backend
from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
from gmpnp_amd.params import pore_parameters, utilities_dir
from gmpnp_amd.problem import pore_problem
levels = [int(x) for x in sys.argv[1:]] or [0, 1, 2]
pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
for lev in levels:
    t0 = time.time()
    prob, _ = pore_problem(pp, mesh, refine=lev)
    nv = prob.coords.shape[0]
    dev = backend.DeviceSolver(prob)
    t1 = time.time()
    dev.set_state(np.zeros(prob.ndof), np.tile(np.r_[np.ones(8), 0.0], nv))
    opts = backend.newton_options({"newton_solver": {"linear_solver": "mumps", "maximum_iterations": 50, "relative_tolerance": 1e-4,
                                                     "absolute_tolerance": 1e-4, "relaxation_parameter": 0.9}})
    t2 = time.time(); st = dev.newton_solve(opts); t3 = time.time()
    nb, nd = dev.n_blocks, dev.ndof
    alg = 648 * nb + 4 * nb + 4 * (nv + 1) + 16 * nd
    print("level %d: %d vertices, %d cells, %d dofs, %.1f MB Jacobian; setup %.1fs; newton %d its, krylov %s, %.3f s -> %.1f its/s"
          % (lev, nv, len(prob.cells), nd, 648e-6 * nb, t1 - t0, st["iterations"], st["krylov_per_iteration"], t3 - t2, st["iterations"] / (t3 - t2)), flush=True)
    for k, name in ((0, "spmv_plain"), (4, "bicg_a"), (5, "bicg_b"), (14, "bicg_a_mat"), (15, "bicg_b_mat"), (16, "vec_a"), (17, "vec_b"), (6, "coarse_a"),
                    (7, "coarse_b"), (1, "element"), (2, "jac_gather"), (3, "res_gather")):
        us = dev.time_kernel(k, 50)
        extra = "  %.0f GB/s algorithmic" % (alg / us / 1e3) if k in (0, 4, 5, 14, 15) else ""
        print("   %-12s %9.2f us%s" % (name, us, extra), flush=True)
    dev.close()
