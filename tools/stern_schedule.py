"""The reference's own 1D schedules, run on the GPU, against the only hot-path outputs the reference holds.

1D/Stern_CO2ER.py:66-68 records field_OHP [V/nm] and eps_rel_OHP "obtained from solving the MPNP code" at
voltage_multiplier = -2.5 ... -12.5 (the 1D defaults otherwise: K+, 0.1 M KHCO3, MPNP, 50 um mesh).  The run length
behind those numbers is not stated.  This tool runs both schedules the script has (1D:256-290):

  dry run   100 steps of 1e-5 s                                   (the only one reachable from the CLI)
  staged    10,000 steps of 1e-5 s, then 10,000 steps at which the clock advances by 1e-3 s while the FORM keeps
            dt = 1e-5 s (the Python name is rebound, the Constant inside F is not: SURVEY Q2) -> 20,000 solves

and prints, per voltage, the driver's field_OHP / eps_rel_OHP (EDLRun.ohp_summary: the reference's projection and
rescaling, 1D:802-805,893-954) at checkpoints with their deviation from the recorded digits.  JSON goes to argv[1].

    python tools/stern_schedule.py profiles/r02/stern_schedule.json [max_steps]
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gmpnp_amd.edl1d import EDLRun

REC = {-2.5: (-0.08032108300135771, 74.56149297894756), -5.0: (-0.2524415478848975, 57.64572780716129),
       -7.5: (-0.4612956299192668, 50.16243860179017), -10.0: (-0.6149631587776277, 49.311548142969336),
       -12.5: (-0.7310301485096051, 49.2556833480052)}
CHECK = (100, 300, 1000, 3000, 10000, 20000)
out_path = sys.argv[1] if len(sys.argv) > 1 else None
max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
table = []
for V, (E, eps) in REC.items():
    t0 = time.perf_counter()
    run = EDLRun(voltage_multiplier=V, dry_run=False)   # staged schedule, Q2 semantics
    assert run.tot_num_steps == 20000
    row = {"voltage_multiplier": V, "recorded_field_OHP": E, "recorded_eps_rel_OHP": eps, "checkpoints": []}
    try:
        for n in range(1, max_steps + 1):
            run.step(verbose=False)
            run.history = run.history[-1:]   # the reference keeps every step (O(n^2) vstack); only the last one is needed here
            if n in CHECK:
                s = run.ohp_summary()
                row["checkpoints"].append({"steps": n, "field_OHP": s["field_OHP"], "eps_rel_OHP": s["eps_rel_OHP"],
                                           "field_rel_dev": s["field_OHP"] / E - 1.0, "eps_rel_dev": s["eps_rel_OHP"] / eps - 1.0,
                                           "newton_iterations_so_far": int(sum(run.newton_its))})
                print("V %6.1f  step %6d  field %.6f (rec %.6f, %+.3f %%)  eps %.4f (rec %.4f, %+.3f %%)  newton %d  %.1fs"
                      % (V, n, s["field_OHP"], E, 100 * (s["field_OHP"] / E - 1), s["eps_rel_OHP"], eps, 100 * (s["eps_rel_OHP"] / eps - 1),
                         sum(run.newton_its), time.perf_counter() - t0), flush=True)
    except Exception as e:  # noqa: BLE001   (a Newton failure is a finding, not a crash of the tool)
        row["error"] = "%s at step %d" % (e, run.n)
        print("V %6.1f  stopped: %s" % (V, row["error"]), flush=True)
    finally:
        run.sys.close()
    row["seconds"] = time.perf_counter() - t0
    table.append(row)
if out_path:
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump({"source": "tools/stern_schedule.py on 1 x MI355X", "rows": table}, fh, indent=1)
