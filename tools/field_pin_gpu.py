"""End-to-end pin against the only hot-path outputs the reference holds: field_OHP [V/nm] and eps_rel_OHP recorded in
1D/Stern_CO2ER.py:66-68 for voltage_multiplier = -2.5 ... -12.5 (K+, 0.1 M KHCO3, MPNP, 50 um mesh = the 1D defaults).
Runs the 1D driver on the GPU until the OHP field has settled and prints model vs recorded."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gmpnp_amd.edl1d import EDLRun
REC = {-2.5: (-0.08032108300135771, 74.56149297894756), -5.0: (-0.2524415478848975, 57.64572780716129),
       -7.5: (-0.4612956299192668, 50.16243860179017), -10.0: (-0.6149631587776277, 49.311548142969336),
       -12.5: (-0.7310301485096051, 49.2556833480052)}
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for V, (E, eps) in REC.items():
    t0 = time.perf_counter()
    run = EDLRun(num_steps=nsteps, voltage_multiplier=V)
    ep, mesh = run.ep, run.mesh
    i0 = int(np.argmin(mesh.coords[:, 0]))
    out = []
    for n in range(nsteps):
        run.step(verbose=False)
        if (n + 1) % (nsteps // 8) == 0:
            p = run.history[-1][:, 6]
            fld = run.sys.dev.project_gradient(p, sign=-1.0)[:, 0]
            out.append(fld[i0] * ep.thermal_voltage / ep.L_n * 1e-9)
    print("V %6.1f recorded %.5f | field_OHP at steps n/8..n: %s | last/recorded %.4f | %.1fs" % (V, E, " ".join("%.5f" % o for o in out), out[-1] / E, time.perf_counter() - t0), flush=True)
    run.sys.close()
