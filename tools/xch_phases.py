"""Development probe: phase timestamps inside the exchange-prologue launch k_half_a_x (peer transport, ONE rank = the whole mesh as one
partition).  Needs a library built with -DGMPNP_XTIMING -DGMPNP_DEV_HOOKS, passed via GMPNP_LIB (tools/xch_phases.sh builds it)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch.distributed as tdist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
tdist.init_process_group("gloo", rank=0, world_size=1)
from gmpnp_amd import backend, dist
from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
from gmpnp_amd.params import pore_parameters, utilities_dir
from gmpnp_amd.problem import pore_problem
pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
prob, _ = pore_problem(pp, mesh)
nv = mesh.num_vertices
opts = backend.newton_options({"nonlinear_solver": "newton", "newton_solver": {"linear_solver": "mumps", "maximum_iterations": 50, "relative_tolerance": 1e-4,
                                                                                "absolute_tolerance": 1e-4, "relaxation_parameter": 0.9}})
NAMES = {0: "X0 exchange wg 0 enters", 1: "X1 wg 0 wave 0: partials summed", 2: "X2 wg 0: flagged words stored (not drained)",
         16: "C0 coarse wg 0 enters", 17: "C1 coarse: loads issued, starts polling", 18: "C2 coarse: every word arrived, sums formed (barrier)", 19: "C3 coarse: scalars",
         20: "C4 coarse: ticket published", 24: "T0 first tile enters", 25: "T1 tile: requests issued", 26: "T2 tile: hand-over", 28: "T3 tile: x staged", 27: "T4 tile: done"}
with dist.PartitionedSolver(prob, 1, rank=0, transport="peer") as ps:
    print("exchange form", ps.exchange_form())
    ps.set_state(np.zeros(nv * 9), np.tile(np.r_[np.ones(8), 0.0], nv))
    st = ps.newton_solve(opts)
    buf = np.zeros(140 * 32)
    lib = ps.lib
    lib.gmpnp_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
    lib.gmpnp_debug_read(ps.devs[0]._h, 14, buf.ctypes.data, buf.size)
    s = buf[140 * 16 + 32: 140 * 16 + 64]
    t0 = s[0]
    for i in sorted(NAMES, key=lambda i: s[i]):
        if s[i] > 0:
            print("%7.2f us  %s" % ((s[i] - t0) / 100.0, NAMES[i]))
tdist.destroy_process_group()
