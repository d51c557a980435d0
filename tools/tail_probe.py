import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
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
dev.linear_solve(F)
for rep in range(3):
    print(os.environ.get("GMPNP_LIB", "default").split("/")[-1], "k_half_a %.2f us  k_half_b %.2f us" % (dev.time_kernel(12, 400), dev.time_kernel(13, 400)))
