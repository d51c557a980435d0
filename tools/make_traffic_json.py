"""Combine two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 3 --warmup 1 --no-cpu-baseline` into
profiles/<round>/bench_steps3_pmc_fetch_write.json (all kernels) and profiles/spmv_pmc.json (the roofline kernels;
bench.py reads `hbm_bytes_per_launch` from it and DROPS it when `build_id` is not the loaded library's).

  (GPU box)  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
             rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
             python tools/pmc_summary.py gpurun_out/pmc_fetch > gpurun_out/pmc_fetch.json
             python tools/pmc_summary.py gpurun_out/pmc_write > gpurun_out/pmc_write.json
  (here)     python tools/make_traffic_json.py gpurun_out/pmc_fetch.json gpurun_out/pmc_write.json r02
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
RND = sys.argv[3] if len(sys.argv) > 3 else "r02"
fetch = json.load(open(sys.argv[1]))
write = json.load(open(sys.argv[2]))
out = {"FETCH_SIZE": {}, "WRITE_SIZE": {}}
for name, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
    for k, cs in d.items():
        if name in cs:
            out[name][k] = {"launches": cs[name]["launches"], "median_KB": cs[name]["median"], "mean_KB": cs[name]["mean"]}
os.makedirs(os.path.join(ROOT, "profiles", RND), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", RND, "bench_steps3_pmc_fetch_write.json"), "w"), indent=1)

def per_launch(kernel):
    f = [v for k, v in out["FETCH_SIZE"].items() if kernel in k][0]["mean_KB"]
    w = [v for k, v in out["WRITE_SIZE"].items() if kernel in k][0]["mean_KB"]
    return 2.0 * f * 1024.0 + w * 1024.0

fused = any("k_half_a<9>" in k for k in out["FETCH_SIZE"])  # two-launch form: the coarse workgroups ride inside the tile launches
ka, kb = ("k_half_a<9>", "k_half_b<9>") if fused else ("k_bicg_a<9>", "k_bicg_b<9>")
a, b = per_launch(ka), per_launch(kb)
json.dump({
    "build_id": ge.source_hash(), "launches_per_krylov_iteration": 2 if fused else 4,
    "source": "profiles/%s/bench_steps3_pmc_fetch_write.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 3 --warmup 1 --no-cpu-baseline`)" % RND,
    "correction": "bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024; the factor 2 on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md (HBM / rocprofv3 section), calibrated here for the solver's own access width (tools/calib_fetch.hip: a 2 GiB stream read with 8 B per lane / 63 active lanes reports FETCH_SIZE*1024 = 0.500005 of the bytes, same as 16 B per lane)",
    "kernels": [ka, kb], "a_bytes_per_launch": a, "b_bytes_per_launch": b, "hbm_bytes_per_launch": 0.5 * (a + b),
    "note": "memory-side (L2 miss) traffic; Infinity-Cache hits are included in FETCH_SIZE, so this is an upper bound on HBM bytes for this < 60 MB working set",
}, open(os.path.join(ROOT, "profiles", "spmv_pmc.json"), "w"), indent=1)
print("%s %.2f MB, %s %.2f MB per launch" % (ka, a / 1e6, kb, b / 1e6))
