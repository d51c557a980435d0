"""The reference's staged 1D schedule (20,000 solves) on the C oracle (oracle/edl1d_oracle.c) for one voltage,
with a chosen arithmetic ("double" / "long double") and chosen Gauss rules for F and J.  Prints Newton statistics per
chunk and field_OHP / eps_rel_OHP against the digits recorded in 1D/Stern_CO2ER.py:66-68.

    python tools/oracle_stern_experiment.py V kind nq_f nq_j [max_steps] [out.json]
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import edl1d

V, kind, nq_f, nq_j = float(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
max_steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20000
out = sys.argv[6] if len(sys.argv) > 6 else None
s = edl1d.Setup(voltage_multiplier=V, dry_run=False, nq_f=nq_f, nq_j=nq_j)
assert s.tot_num_steps == 20000
E, eps = edl1d.RECORDED[V]
u, un = s.initial_state()
t0, n, chunk, total_its = time.time(), 0, 250, 0
rows = []
while n < max_steps:
    k = min(chunk, max_steps - n)
    u, un, its, done, res = edl1d.run(s, k, u, un, kind=kind)
    n += done
    good = its[:done]
    total_its += int(good.sum())
    o = s.ohp_summary(un)
    row = {"steps": n, "newton_total": total_its, "its_min": int(good.min()) if done else None, "its_max": int(good.max()) if done else None,
           "its_mean": float(good.mean()) if done else None, "field_OHP": o["field_OHP"], "eps_rel_OHP": o["eps_rel_OHP"],
           "field_rel_dev": o["field_OHP"] / E - 1, "eps_rel_dev": o["eps_rel_OHP"] / eps - 1, "seconds": time.time() - t0}
    rows.append(row)
    print("V %.1f %s F%d/J%d step %5d its min/mean/max %s/%.2f/%s  field %.16g (%+.2e)  eps %.16g (%+.2e)  %.0fs" % (
        V, kind, nq_f, nq_j, n, row["its_min"], row["its_mean"] or 0, row["its_max"], o["field_OHP"], row["field_rel_dev"],
        o["eps_rel_OHP"], row["eps_rel_dev"], row["seconds"]), flush=True)
    if done < k:
        bad = -its[done] - 1
        print("solve %d did not converge: %d iterations, residuals %s" % (n + 1, bad, " ".join("%.3e" % r for r in res[: bad + 1])), flush=True)
        rows.append({"failed_solve": n + 1, "residuals": [float(r) for r in res[: bad + 1]]})
        np.savez("/tmp/exp/fail_V%s_%s_%d%d.npz" % (V, kind.replace(" ", ""), nq_f, nq_j), u=u, un=un, step=n + 1)
        break
if out:
    with open(out, "w") as fh:
        json.dump({"voltage_multiplier": V, "arithmetic": kind, "nq_f": nq_f, "nq_j": nq_j, "recorded": [E, eps], "rows": rows}, fh, indent=1)
