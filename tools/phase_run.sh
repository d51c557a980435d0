export GMPNP_LIB=$PWD/abtest/lib_timing.so
for k in 12 13; do echo "== kernel $k"; python tools/phase_times.py $k; done
python - <<'PY'
import sys; sys.path.insert(0,'.')
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
dev.set_state(u, u); F,_ = dev.assemble(True); dev.linear_solve(F)
for k,name in ((0,'spmv_plain'),(9,'stream_read 2048 WG'),(10,'stream_read 512 WG'),(11,'stream_read 8192 WG'),(4,'bicg_a'),(5,'bicg_b'),(6,'coarse_a'),(7,'coarse_b'),(12,'half_a'),(13,'half_b'),(8,'tiny 1 WG')):
    print('%-22s back-to-back %.2f us' % (name, dev.time_kernel(k, 200)))
PY
