"""Development probe: phase timestamps inside k_bicg_a (needs a library built with -DGMPNP_TIMING -DGMPNP_DEV_HOOKS, passed via GMPNP_LIB)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
which = int(sys.argv[1]) if len(sys.argv) > 1 else 4
print("back-to-back %.2f us" % dev.time_kernel(which, 100))
buf = np.zeros(140 * 32)
dev.lib.gmpnp_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
dev.lib.gmpnp_debug_read(dev._h, 14, buf.ctypes.data, buf.size)
st = buf[140 * 16:140 * 16 + 24].reshape(3, 8)
t00 = st[:, 0].min()
for name, row in zip(("block 0", "middle block", "last block"), st):
    print("%-12s start %+7.2f us | phases (us from tile start): %s" % (name, (row[0] - t00) / 100.0, " ".join("%6.2f" % ((x - row[0]) / 100.0) for x in row[1:6])))
occ = (ctypes.c_int * 4)()
dev.lib.gmpnp_debug_occupancy(occ)
print("resident workgroups/CU: bicg_a %d bicg_b %d spmv_plain %d coarse_a %d; tiles %d" % (occ[0], occ[1], occ[2], occ[3], dev.n_blocks and -(-dev.ndof // 63)))
