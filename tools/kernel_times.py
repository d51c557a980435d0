import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from gmpnp_amd import backend
from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
from gmpnp_amd.params import pore_parameters, utilities_dir
from gmpnp_amd.problem import pore_problem
pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
prob, _ = pore_problem(pp, mesh)
nv = mesh.num_vertices
rng = np.random.default_rng(0)
u = np.concatenate([rng.uniform(.5, 1.5, (nv, 8)), rng.uniform(-1, 0, (nv, 1))], axis=1).ravel()
dev = backend.DeviceSolver(prob)
dev.set_state(u, u)
F, _ = dev.assemble(True)
dev.linear_solve(F)  # builds the preconditioner, leaves Krylov vectors populated
names = {9: "stream 32MB (2048 wg)", 10: "stream 32MB (512 wg)", 11: "stream 32MB (8192 wg)", 0: "spmv_plain", 4: "bicg_a", 5: "bicg_b", 6: "coarse_a", 7: "coarse_b", 8: "tiny copy (1 wave)", 1: "element", 2: "jac_gather", 3: "res_gather"}
for k, n in names.items():
    print("%-20s %.2f us (back-to-back)" % (n, dev.time_kernel(k, 200)))
