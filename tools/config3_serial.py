"""BASELINE configs[3] parameters (L=100 nm, R=50 nm, 1.0 M on the L_10_R_5 geometry) on ONE GPU with the device solver."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gmpnp_amd import backend
from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path, mark_pore_boundaries, pore_wall_tolerance
from gmpnp_amd.params import pore_parameters, utilities_dir
from gmpnp_amd.problem import Problem, pore_dirichlet
from gmpnp_amd.pore3d import SOLVER_PARAMETERS
pp = pore_parameters(concentration_elec=1.0, L=100e-9, R=50e-9)
mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), "L_10_R_5.xml"))
bnd = mark_pore_boundaries(mesh, pp.aspect_pore, pore_wall_tolerance(10e-9, 5e-9))
dofs, vals = pore_dirichlet(pp, bnd)
prob = Problem(coords=mesh.coords, cells=mesh.cells, model=pp.model, wall_facets=bnd.ds_facets[2], exit_facets=bnd.ds_facets[3], bc_dofs=dofs, bc_vals=vals)
nv = mesh.num_vertices
with backend.DeviceSolver(prob) as dev:
    dev.set_state(np.zeros(prob.ndof), np.tile(np.r_[np.ones(8), 0.0], nv))
    t0 = time.perf_counter()
    try:
        st = dev.newton_solve(backend.newton_options(SOLVER_PARAMETERS))
        print("serial device solve: its", st["iterations"], "krylov", st["krylov_per_iteration"][:st["iterations"]], "residuals", ["%.2e" % r for r in st["residuals"]], "%.3f s" % (time.perf_counter() - t0))
    except Exception as e:
        print("serial device solve FAILED:", str(e)[:300])
