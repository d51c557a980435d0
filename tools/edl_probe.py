"""1D EDL case (BASELINE configs[1]: 50 um mesh, K+, 0.1 M, V=-1): Newton iterations/s on the GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from gmpnp_amd.edl1d import EDLRun
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
run = EDLRun(num_steps=steps) if "num_steps" in EDLRun.__init__.__code__.co_varnames else EDLRun()
for _ in range(2): run.step(verbose=False)
t0 = time.perf_counter()
n0 = sum(run.newton_its)
for _ in range(steps): run.step(verbose=False)
dt = time.perf_counter() - t0
its = sum(run.newton_its) - n0
print("1D 50um: %d vertices, %d steps, %d Newton its, %.3f s -> %.1f its/s, %.2f ms/step" % (run.mesh.num_vertices, steps, its, dt, its / dt, 1e3 * dt / steps))
