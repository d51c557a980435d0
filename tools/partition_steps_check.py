"""In-process partitioned time loop (all ranks in one process on one GPU) against the single-GPU run: Newton counts per step and
final state, P = 2, 4, 8 on L_50_R_5 (exercises the breakdown retry of the partitioned BiCGStab over many solves)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gmpnp_amd.pore3d import PoreRun
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
common = dict(num_steps=steps, concentration_elec=0.5, L=50e-9, R=5e-9)
ref = PoreRun(**common)
for _ in range(steps): ref.step(verbose=False)
uref = np.asarray(ref.history[-1]) if hasattr(ref, "history") else None
print("serial Newton", ref.newton_its, "krylov", ref.sys.krylov_iterations)
for P in (2, 4, 8):
    run = PoreRun(partition=(P, None), **common)
    for _ in range(steps): run.step(verbose=False)
    u = np.asarray(run.history[-1])
    err = float(np.abs(u - uref).max() / np.abs(uref).max())
    print("P=%d Newton %s krylov %d  same counts %s  rel.err %.2e" % (P, run.newton_its, run.sys.krylov_iterations, run.newton_its == ref.newton_its, err))
    run.sys.close()
ref.sys.close()
