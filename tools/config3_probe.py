"""BASELINE configs[3]: 3D pore L_100_R_50 (generated mesh, gmpnp_amd.meshgen), 1.0 M KHCO3, serial and 4-way partitioned."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gmpnp_amd.pore3d import PoreRun
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
conc = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
out = {}
for tag, part in (("serial", None), ("4 partitions", (4, None))):
    try:
        t0 = time.perf_counter()
        run = PoreRun(num_steps=nsteps, concentration_elec=conc, L=100e-9, R=50e-9, partition=part)
        t1 = time.perf_counter()
        run.run(verbose=False)
        t2 = time.perf_counter()
        out[tag] = (list(run.newton_its), np.array(run.history[-1]))
        print("%s: %d vertices, create %.1f s, %d steps %.2f s, Newton its %s, Krylov %d, CO2_min %.4f" % (
            tag, run.mesh.num_vertices, t1 - t0, nsteps, t2 - t1, run.newton_its, run.sys.krylov_iterations, run.CO2_min), flush=True)
        run.sys.close()
    except Exception as e:  # noqa: BLE001
        print(tag, "FAILED:", str(e)[:300], flush=True)
if len(out) == 2:
    a, b = out["serial"], out["4 partitions"]
    print("same Newton counts:", a[0] == b[0], " rel. difference of the final states: %.2e" % (np.linalg.norm(a[1] - b[1]) / np.linalg.norm(a[1])))
