"""L_50_R_1 (the thinnest pore of the sweep): Krylov behaviour per Newton iteration for different coarse spaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmpnp_amd.pore3d import PoreRun
for na in (8,):
    run = PoreRun(num_steps=8, concentration_elec=0.5, L=50e-9, R=1e-9, device_kwargs={"n_aggregates": na})
    out = []
    try:
        for n in range(8):
            st = run.step(verbose=False)
            out.append((st["iterations"], list(st.get("krylov_per_iteration", []))[:st["iterations"]]))
        print("nagg", na, "ok", out, flush=True)
    except RuntimeError as e:
        print("nagg", na, "FAILED after", out, str(e)[:150], flush=True)
    run.sys.close()
