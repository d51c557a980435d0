#!/usr/bin/env python3
"""Headline benchmark: Newton iterations / second (+ achieved HBM GB/s of the dominant kernel) of the 3D pore
case of BASELINE.json (configs[2]: MPNP_CO2ER_pore, L_50_R_5.xml, 0.5 M KHCO3, V = -1) on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one time step of the reference's loop (3D/MPNP_CO2ER_pore.py:783-858): a damped Newton solve on the
device (residual/Jacobian assembly, two-level BiCGStab, update) plus the per-step host glue (vertex values back,
median -> Sechenov -> new Dirichlet value, u_n.assign(u)).  W warm-up steps run first and are discarded (state reset to
t = 0), then EXACTLY K steps from t = 0 are timed between barrier + synchronize pairs; value = Newton iterations of
all ranks / max-over-ranks time.

N > 1: `value` is the north-star quantity — ONE L_50_R_5 problem, mesh-partitioned over the N GPUs (z-slab partitions,
one process per GPU; ghost-row exchange and one fused all-reduce per BiCGStab half-iteration inside libgmpnp.so, global
coarse space): strong scaling, Newton iterations of the one problem / max-over-ranks time.  Transports in order of preference:
peer mailboxes over xGMI, RCCL, host-staged; the first that works is `value` (GMPNP_BENCH_ALL_TRANSPORTS=1 times both device
transports and takes the faster, `partitioned.transports_timed`).  On a
3.7k-vertex mesh that cannot beat one GPU (a half-iteration is 10 us of kernel against two collectives); `--refine 1|2`
gives the sizes where it can.  The same invocation first times N independent replicas of the problem, one per GPU (the
parameter-sweep mapping of BASELINE configs[4], no collective, weak scaling) and reports them under `replicas`; should the
partitioned phase fail or hang on the node (its RCCL path cannot be rehearsed on a one-GPU box), the replica figure is
what `value` falls back to, with the reason under `partitioned`.

Extra objects on the JSON line:
  roofline      dominant kernels = the two fused BiCGStab half-iterations (k_half_a / k_half_b on meshes whose workgroups are
                all resident at once, else k_bicg_a / k_bicg_b after a separate coarse launch): one SELL block SpMV each
                plus the vector updates.  achieved = algorithmic bytes of ONE SpMV (648 nb + 4 nb + 4 (nv+1) +
                16 nd; the vector traffic fused in is not counted) / mean duration of a half-iteration, measured LIVE
                during the timed region: the first burst of every 4th linear solve — 40-odd back-to-back half-iterations,
                all of them live — sits between ONE pair of HIP events on the solver's stream; elapsed time / number of
                half-iterations = the launch plus the gap to the next one (in the forms with a separate coarse launch, that
                launch too).  rocprofv3's kernel durations of the same command (profiles/) are the check.
  cpu_baseline  the CPU oracle (NumPy assembly + SciPy SuperLU) timed on rank 0 / N = 1 over the first 3 Newton iterations
                of the same window, on 1 thread and on all cores (about 35 s together); kind = "port" (FEniCS itself
                cannot be installed).  `full_window_recorded` = the committed timing of the whole 50-step window.
                `banded_all_cores`: the same Jacobian in the library's slab order is a band of ~1,650 scalars; one LAPACK
                dgbsv (scipy.linalg.solve_banded, threaded BLAS) on every core of the box, and under `block_band_openmp` the
                same band as 9x9 node blocks factored by oracle/band_lu_omp.c with OpenMP (the leg that does scale with
                cores) — the stand-ins for the reference's MUMPS (3D:792) that use the box.  `value` is the FASTEST of the
                legs (`cores` = the threads it used).
  edl50         (N = 1) BASELINE configs[1]: the 1D script's 100 dry-run steps (50 um mesh, 7 fields, 41,937 dofs; reference
                1D:256-268) — Newton iterations / s, the roofline of its direct solve (block cyclic reduction, k_bcr_*), and the
                C oracle timed on the same 100 steps.  `--case edl50` makes that the headline line instead.

Exit codes: 0 = a line was printed and every phase finished; 3 = the partitioned phase hung and the watchdog printed the line
(with what was measured before) — the driver sees the hang.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--mesh", type=str, default="L_50_R_5", help="L_<nm>_R_<nm> pore mesh (default: the north-star mesh)")
    p.add_argument("--refine", type=int, default=0, help="uniform refinements of the mesh (0 = the reference mesh itself)")
    p.add_argument("--replicas-only", action="store_true", help="N > 1: time only the N independent replicas (no partitioned solve)")
    p.add_argument("--force-partitioned", action="store_true",
                   help="also at N = 1: run the partitioned phase (one partition, RCCL communicator of one rank) — rehearsal of "
                        "the N > 1 code path on a one-GPU box; reported under `partitioned_rehearsal`, `value` stays the single-GPU solver's")
    p.add_argument("--partition-timeout", type=int, default=240, help="seconds the partitioned phase may take before the replica result is reported alone")
    p.add_argument("--case", choices=["pore50", "edl50"], default="pore50",
                   help="pore50 = BASELINE configs[2] (the headline); edl50 = BASELINE configs[1], the 1D script's 100 dry-run steps")
    p.add_argument("--no-edl50", action="store_true", help="N = 1: skip the secondary 1D measurement")
    p.add_argument("--multilevel", action="store_true",
                   help="with --refine R > 0: the geometric multilevel term of the preconditioner over the nested meshes (gmpnp_attach_coarse_level); "
                        "same Newton iterates, 8x fewer BiCGStab iterations at R = 2.  On by itself from R = 2 on at N = 1 (where it is 3.7x faster); "
                        "--no-multilevel keeps the two-level scheme")
    p.add_argument("--no-multilevel", action="store_true")
    return p.parse_args()


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CFS quota (cpu.max = "quota period").
    On the one-GPU test box os.cpu_count() says 256 and the quota is 16: a thread pool sized by cpu_count() spends the quota of
    every 100 ms period in its first few ms and is throttled for the rest (the round-2 "all cores is no faster" finding was that)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                tok = fh.read().split()
            if path.endswith("cpu.max"):
                if tok[0] != "max":
                    n = min(n, max(1, int(float(tok[0]) / float(tok[1]))))
            else:
                q = int(tok[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                        n = min(n, max(1, q // int(fh.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(run, max_newton=3):
    """Bounded sample of the CPU path: the first `max_newton` Newton iterations of time step 0 of
    the SAME window with the oracle (test infrastructure, used here only as the timed CPU leg) — NumPy P1 assembly of the
    exact Jacobian + SciPy SuperLU per iteration, damped update — once limited to 1 thread (the reference is a serial
    process) and once with every core of the box (threadpoolctl; SuperLU itself is serial).  The whole 50-step window
    takes the oracle about half an hour: tools/cpu_window.py times it, profiles/r02/cpu_window_*.json hold the result,
    and tests/test_gpu_parity.py::test_bench_window_matches_golden pins the GPU's Newton count per step to the oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gmpnp_oracle as O
    from threadpoolctl import threadpool_limits
    import copy
    prob = copy.copy(run.problem)
    from gmpnp_amd.problem import pore_dirichlet
    prob.bc_dofs, prob.bc_vals = pore_dirichlet(run.pp, run.bnd)   # the t = 0 Dirichlet set
    nv = run.mesh.num_vertices
    u0 = np.zeros(prob.ndof)
    un = np.tile(np.r_[np.ones(8), 0.0], nv)
    O.assemble(prob, u0, un)  # builds the scatter pattern once (one-off set-up, like DOLFIN's sparsity pattern)
    legs = {}
    for name, lim in (("one_thread", 1), ("all_cores", usable_cpus())):
        with threadpool_limits(limits=lim):
            t0 = time.perf_counter()
            _, st = O.newton_solve(prob, u0, un, maximum_iterations=max_newton, relaxation_parameter=0.9,
                                   error_on_nonconvergence=False)
            wall = time.perf_counter() - t0
        legs[name] = {"threads": lim, "newton_iterations": st.iterations, "seconds": wall, "assembly_seconds": st.t_assemble,
                      "lu_seconds": st.t_linear, "value": st.iterations / wall}
    one = legs["one_thread"]
    banded = None
    try:
        banded = banded_leg(O, prob, u0, un)
    except Exception as e:  # noqa: BLE001   (a missing LAPACK binding must not cost the bench line)
        banded = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    window = None
    wpath = os.path.join(ROOT, "profiles", "r02", "cpu_window_1thread.json")
    if os.path.exists(wpath):
        with open(wpath) as fh:
            window = json.load(fh)
    out = {"value": one["value"], "unit": "Newton-iterations/s", "cores": 1, "kind": "port",
           "sample": "first %d Newton iterations of time step 0 of the same window (same mesh/parameters, zero initial guess): "
                     "NumPy P1 assembly of J and F %.1f s + SciPy SuperLU factor+solve %.1f s on 1 thread; FEniCS/MUMPS "
                     "itself is not installable on this box" % (one["newton_iterations"], one["assembly_seconds"], one["lu_seconds"]),
           "all_cores": legs["all_cores"], "one_thread": one, "host_cpus": os.cpu_count(), "usable_cpus": usable_cpus(), "banded_all_cores": banded,
           "full_window_recorded": window}
    omp = (banded or {}).get("block_band_openmp") or {}
    if omp.get("value", 0.0) > out["value"]:
        # the fastest CPU leg is the baseline: one Newton iteration = NumPy assembly (serial) + threaded node-block band LU
        out.update(value=omp["value"], cores=omp["threads"],
                   sample="one Newton iteration of time step 0 of the same window (same mesh/parameters, zero initial guess): NumPy P1 "
                          "assembly of J and F %.2f s (1 thread) + node-block band LU of the slab-ordered Jacobian with OpenMP on %d "
                          "threads %.2f s (oracle/band_lu_omp.c; answer = SuperLU's to %.0e); the serial SuperLU leg (%.3f its/s) is "
                          "`one_thread`; FEniCS/MUMPS itself is not installable on this box"
                          % (banded["assembly_seconds"], omp["threads"], omp["lu_seconds"], omp["solution_vs_superlu"], one["value"]))
    return out


def banded_leg(O, prob, u0, un):
    """One Newton iteration's linear algebra the way a band solver does it, on every core: the Jacobian of time step 0 in the
    library's slab order (vertices sorted along the pore axis: half-bandwidth ~183 node blocks = ~1,650 scalars on L_50_R_5)
    factored and solved by LAPACK dgbsv through scipy.linalg.solve_banded with the BLAS threads the box offers.  The answer is
    checked against SuperLU's.  Reference: MUMPS (3D:792), which is multi-frontal and threaded; this is the stand-in that uses the
    box's cores.  Capped at 4 GiB of band storage."""
    import scipy.linalg as sla
    import scipy.sparse.linalg as spla
    from threadpoolctl import threadpool_limits
    from gmpnp_amd.backend import slab_permutation
    t0 = time.perf_counter()
    b, A = O.assemble(prob, u0, un)
    t_asm = time.perf_counter() - t0
    nf, nv = prob.nf, prob.coords.shape[0]
    perm = slab_permutation(prob.coords, prob.cells, window=0)      # internal position -> file vertex
    pos = np.empty(nv, dtype=np.int64)
    pos[perm] = np.arange(nv)
    dof_new = (pos[:, None] * nf + np.arange(nf)[None, :]).ravel()   # file dof -> band dof
    C = A.tocoo()
    r, c = dof_new[C.row], dof_new[C.col]
    kl = int((r - c).max())
    ku = int((c - r).max())
    n = A.shape[0]
    if (2 * kl + ku + 1) * n * 8 > 4 * 2 ** 30:
        return {"skipped": "band storage %.1f GiB" % ((2 * kl + ku + 1) * n * 8 / 2 ** 30)}
    ab = np.zeros((kl + ku + 1, n))
    ab[ku + r - c, c] = C.data
    rhs = np.empty(n)
    rhs[dof_new] = b
    threads = usable_cpus()
    with threadpool_limits(limits=threads):
        t0 = time.perf_counter()
        x = sla.solve_banded((kl, ku), ab, rhs, overwrite_ab=True, overwrite_b=False, check_finite=False)
        t_lu = time.perf_counter() - t0
    xs = spla.splu(A.tocsc()).solve(b)
    err = float(np.linalg.norm(x[dof_new] - xs) / np.linalg.norm(xs))
    out = {"threads": threads, "half_bandwidth_scalars": [kl, ku], "assembly_seconds": t_asm, "lu_seconds": t_lu,
           "seconds_per_newton_iteration": t_asm + t_lu, "value": 1.0 / (t_asm + t_lu), "unit": "Newton-iterations/s",
           "solution_vs_superlu": err,
           "what": "Jacobian of time step 0, slab order, LAPACK dgbsv (scipy.linalg.solve_banded) with %d BLAS threads" % threads}
    del ab
    if nf == 9:
        # the same band as NODE BLOCKS, factored by oracle/band_lu_omp.c with OpenMP over the window behind each pivot: the leg
        # that does get faster with cores (dgbsv's rank-1 panel updates do not)
        try:
            import band_lu
            scaling = {}
            counts = sorted({min(8, threads), threads})
            for t in counts:
                xo, dt, hb = band_lu.solve(A, b, pos, threads=t)
                scaling[str(t)] = dt
            erro = float(np.linalg.norm(xo - xs) / np.linalg.norm(xs))
            out["block_band_openmp"] = {"threads": threads, "half_bandwidth_blocks": hb, "lu_seconds": scaling[str(threads)],
                                        "lu_seconds_by_threads": scaling, "solution_vs_superlu": erro,
                                        "seconds_per_newton_iteration": t_asm + scaling[str(threads)],
                                        "value": 1.0 / (t_asm + scaling[str(threads)]), "unit": "Newton-iterations/s",
                                        "what": "the same system as node-block band LU (oracle/band_lu_omp.c, gcc -fopenmp), the algorithm of "
                                                "the library's own direct fallback on the host"}
        except Exception as e:  # noqa: BLE001
            out["block_band_openmp"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return out


def edl50_case(device_id, steps=100, warmup=3, cpu=True, repeats=1):
    """BASELINE configs[1]: reference 1D/MPNP_CO2ER_EDL.py with its defaults (K+, 0.1 M KHCO3, MPNP, V = -1, 50 um mesh of 5,991
    vertices, 7 fields = 41,937 dofs), the 100 dry-run steps of 1D:256-268.  A step = one Newton solve on the device (element
    pass, gathers, block-cyclic-reduction direct solve, update) + the step glue (vertex values back, u_n.assign(u))."""
    import torch
    from gmpnp_amd.edl1d import EDLRun
    run = EDLRun(device_kwargs={"device_id": device_id})
    try:
        assert run.tot_num_steps == 100 and run.mesh.num_vertices == 5991
        nv = run.mesh.num_vertices

        def reset():
            run.sys.initialise([1.0] * 6 + [0.0])
            run.history = run.history[:1]
            run.newton_its, run.n, run.t = [], 0, 0.0

        for _ in range(warmup):
            run.step(verbose=False)
        # `repeats` > 1 (the secondary measurement inside the 3D line): the window is 50-odd ms of host-paced launches, and one
        # hiccup of a shared host halves its rate (seen: 1,702 against 3,816 its/s in two consecutive runs) — the MEDIAN window
        # is reported there, all of them listed; the headline form (--case edl50) times its K steps once, as the contract says
        windows = []
        for _ in range(max(1, repeats)):
            reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                run.step(verbose=False)
            torch.cuda.synchronize()
            windows.append(time.perf_counter() - t0)
        dt = sorted(windows)[len(windows) // 2]
        its = int(sum(run.newton_its))
        dev = run.sys.dev
        nf = dev.nf
        nd = dev.ndof
        # direct solve of the block-tridiagonal Jacobian: every matrix block read once, right-hand side in, solution out
        alg_bytes = (3 * nv - 2) * nf * nf * 8 + 2 * 8 * nd
        solve_us = dev.time_kernel(18, 50)
        levels = int(np.ceil(np.log2(nv))) + 1
        out = {"metric": "newton_iterations_per_sec", "value": its / dt, "unit": "Newton-iterations/s", "steps": steps, "warmup": warmup,
               "ms_per_step": 1e3 * dt / steps, "dtype": "f64",
               "config": {"workload": "1D MPNP_CO2ER_EDL, 1D_variable_50um_mesh_5990, K+, 0.1 M KHCO3, V=-1: the %d dry-run steps "
                                      "(Newton rtol=atol=1e-4, omega=1, max 50; linear solve = block cyclic reduction, direct)" % steps,
                          "n_vertices": nv, "n_dofs": nd, "newton_iterations": its, "windows_seconds": windows},
               "roofline": {"bound": "hbm", "kernel": "1D direct solve, block cyclic reduction over %d levels: k_tri_extract, k_bcr_forward per level, k_bcr_tail "
                                                      "(the levels of up to 4 rows, the single row, and back: one wave), k_bcr_backward per level" % (levels - 1),
                            "algorithmic_bytes_per_solve": alg_bytes, "mean_solve_us": solve_us,
                            "achieved": alg_bytes / (solve_us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                            "frac": alg_bytes / (solve_us * 1e-6) / 1e9 / 8000.0, "traffic": None,
                            "note": "a chain of ~20 dependent launches over 7.7 MB that halves at every level, each a pivoted 7x7 solve by "
                                    "shuffles inside 8-lane groups: latency and LDS-crossbar bound, not bandwidth bound; timed with one HIP event "
                                    "pair around 50 back-to-back solves"}}
    finally:
        run.sys.close()
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import edl1d as OE
        st = OE.Setup()
        u, un = st.initial_state()
        OE.run(st, 1, u, un)   # load / page in
        t0 = time.perf_counter()
        u, un, oits, done, _ = OE.run(st, steps, u, un)
        wall = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": float(oits[:done].sum()) / wall, "unit": "Newton-iterations/s", "cores": 1, "kind": "port",
                               "sample": "the same %d steps with oracle/edl1d_oracle.c (literal Gauss-point assembly + banded LU with "
                                         "partial pivoting, one thread): %d Newton iterations in %.2f s" % (steps, int(oits[:done].sum()), wall),
                               "newton_iterations": int(oits[:done].sum())}
        out["newton_iterations_equal_cpu"] = bool(int(oits[:done].sum()) == its)
    return out


def main():
    a = parse()
    if a.refine >= 2 and not a.no_multilevel and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        a.multilevel = True   # (partitioned handles have no multilevel term yet: N > 1 stays two-level)
    if a.no_multilevel:
        a.multilevel = False
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # rehearsal on a one-GPU box: GMPNP_BENCH_BACKEND=gloo lets several ranks share the card (ranks map onto the visible
    # devices modulo their count); the driver's runs use RCCL ("nccl") with one rank per GPU
    backend = os.environ.get("GMPNP_BENCH_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    # several PROCESSES on one card (rehearsal only; the contract is one rank per GPU): a second hardware queue per
    # process makes the processes time-slice 3x slower (measured 290 vs 857 its/s for two ranks) and an in-launch
    # hand-over must not wait on workgroups another process keeps off the machine: gmpnp_options_t.shared_device
    shared = int(os.environ.get("LOCAL_WORLD_SIZE", world)) > ndev
    local = local % ndev
    if world > 1 or a.force_partitioned:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    if a.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (a.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    from gmpnp_amd.pore3d import PoreRun
    from gmpnp_amd.problem import pore_dirichlet

    if a.case == "edl50":   # BASELINE configs[1] as the headline line: N independent replicas of the 1D run (the 1D path does not shard)
        if dist is not None:
            dist.barrier()
        o = edl50_case(local, steps=a.steps if a.steps != 50 else 100, warmup=a.warmup, cpu=(rank == 0 and not a.no_cpu_baseline))
        val, ms = o["value"], o["ms_per_step"]
        if dist is not None:
            t = torch.tensor([ms], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t[0])
            val = world * o["config"]["newton_iterations"] / (ms * 1e-3 * o["steps"])
        if rank == 0:
            o.update(value=val, ms_per_step=ms, n_gpus=world, higher_is_better=True, scaling="weak", vs_baseline=None,
                     data="reference inputs shipped in data/utilities (1D_variable_50um_mesh_5990, parameters.yaml, bulk_soln_0.1KHCO3.yaml); deterministic, no RNG")
            o["config"]["parallelism"] = "1 GPU" if world == 1 else "%d independent replicas (the 1D path does not shard: replicas only)" % world
            print(json.dumps(o), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    _, Lnm, _, Rnm = a.mesh.split("_")
    common = dict(num_steps=a.steps, concentration_elec=0.5, L=float(Lnm) * 1e-9, R=float(Rnm) * 1e-9, refine=a.refine)
    run = PoreRun(device_kwargs={"device_id": local, "shared_device": int(shared),
                                 "profile_every": int(os.environ.get("GMPNP_BENCH_SAMPLE_EVERY", "4"))},
                  multilevel=bool(a.multilevel and a.refine > 0), **common)
    nv = run.mesh.num_vertices

    def reset(r):
        r.sys.set_bcs(*pore_dirichlet(r.pp, r.bnd))
        r.sys.initialise([1.0] * 8 + [0.0])
        r.history = r.history[:1]
        r.newton_its, r.n, r.t = [], 0, 0.0
        r.sys.krylov_iterations = 0

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(r):
        """W untimed warm-up steps, state back to t = 0, then EXACTLY K steps between barrier + synchronize pairs."""
        for _ in range(a.warmup):
            r.step(verbose=False)
        reset(r)
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            r.step(verbose=False)
        fence()
        return time.perf_counter() - t0

    run.sys.dev.spmv_profile()  # clear the sampler
    for _ in range(a.warmup):
        run.step(verbose=False)
    reset(run)
    run.sys.dev.spmv_profile()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run.step(verbose=False)
    fence()
    dt = time.perf_counter() - t0

    its = float(sum(run.newton_its))
    kry = float(run.sys.krylov_iterations)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        s = torch.tensor([its, kry], dtype=torch.float64, device=red_dev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dt, its, kry = float(t[0]), float(s[0]), float(s[1])

    prof = run.sys.dev.spmv_profile() if rank == 0 else None
    final_replica = run.history[-1].copy()      # what a partitioned run of the same window must reproduce

    # The complete line is built BEFORE the partitioned phase (roofline sampling, event overhead, CPU baseline, 1D case): from
    # here on the only thing that can still change is `partitioned`, so the watchdog has nothing to compute on a GPU that may hang.
    out = make_output(a, run, run.sys.dev, prof, nv, world, dt, its, kry) if rank == 0 else None
    if rank == 0 and world == 1 and not a.no_edl50 and a.refine == 0:
        try:
            out["edl50"] = edl50_case(local, cpu=not a.no_cpu_baseline, repeats=3)
        except Exception as e:  # noqa: BLE001
            out["edl50"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    import threading
    emit_lock = threading.Lock()
    emitted = {"done": False}

    def emit(part_result):
        """Print THE line, once (main path and watchdog share the flag)."""
        with emit_lock:
            if emitted["done"] or rank != 0:
                return
            emitted["done"] = True
            print(json.dumps(attach_partitioned(out, a, world, part_result)), flush=True)

    # ---- N > 1: the north-star quantity — ONE problem, mesh-partitioned over the N GPUs -------------------------------------
    part = None
    if (world > 1 or a.force_partitioned) and not a.replicas_only:
        # the replica handle is idle from here on: without its side stream and hand-over flags nothing of it shares the card
        state = {"best": None, "checks": {}, "phase": "start"}

        def bail():
            # A transport that hangs between physical GPUs cannot be rehearsed on a one-GPU box.  Report what was measured before
            # it (another transport's rate, or the replica measurement alone), which transport and phase hung, and exit NON-ZERO:
            # the line is complete, and the driver must see that the run did not finish.
            best = state["best"]
            info = dict(best) if best is not None else {"error": "partitioned phase did not finish within %d s" % a.partition_timeout}
            info["watchdog"] = "hung in %s; transport checks so far: %s" % (state["phase"], json.dumps(state["checks"]))
            info["transport_checks"] = state["checks"]
            emit(info)
            sys.stdout.flush()
            os._exit(3)

        wd = threading.Timer(a.partition_timeout, bail)
        wd.daemon = True
        wd.start()
        # Transports in order of preference: peer mailboxes (one kernel launch per collective: stores into the other ranks'
        # IPC-mapped mailboxes, xGMI between GPUs); RCCL inside the library; the library's host-staged transport over
        # torch.distributed/gloo (PCIe per collective) — same algorithm in all three.  Every candidate first passes
        # gmpnp_group_selftest (self-checking all-reduce + ghost-row messages over ITS transport, `transport_checks`); its timed run
        # must then take the Newton iterations of the single-GPU run AND end on the single-GPU run's state (1e-8) — a transport
        # that delivers a stale ghost row does not get to report a rate.  The first one that passes is the headline;
        # GMPNP_BENCH_ALL_TRANSPORTS=1 times both device transports and reports the faster one.
        errors, tried = [], []
        # "peer" = the mailboxes with the exchange of a half-iteration riding inside the next launch as flagged words (round 3; like
        # everything about this transport it has only ever run between processes on ONE card), "peer-separate" = the same mailboxes with
        # the flag-based exchange launches of round 2: tried next if the first does not pass, before RCCL
        order = ["peer", "peer-separate"] + (["rccl"] if backend == "nccl" else []) + ["host"]
        first_only = os.environ.get("GMPNP_BENCH_ALL_TRANSPORTS", "0") in ("", "0")
        if os.environ.get("GMPNP_BENCH_TRANSPORTS"):   # rehearsal / comparison runs: e.g. "host" or "rccl,host"
            order = [x for x in os.environ["GMPNP_BENCH_TRANSPORTS"].split(",") if x in ("peer", "peer-separate", "rccl", "host")]
            first_only = True

        def agree(flag):
            """max over ranks of an error flag: every rank takes the same branch"""
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return bool(int(t[0]))

        for transport in order:
            if state["best"] is not None and (first_only or transport == "host" or transport == "peer-separate"):
                if transport == "peer-separate" and not first_only:
                    continue      # (the comparison of GMPNP_BENCH_ALL_TRANSPORTS is between peer and RCCL)
                break
            check = {"selftest": None, "run": None}
            state["checks"][transport] = check
            prun = None
            try:
                state["phase"] = "%s: set-up" % transport
                err = None
                try:
                    dk = {"device_id": local, "transport": "peer" if transport == "peer-separate" else transport, "shared_device": int(shared)}
                    if transport == "peer-separate":
                        dk["exchange_form"] = 1
                    elif transport == "peer" and os.environ.get("GMPNP_BENCH_EXCHANGE_FORM"):   # A/B runs: 1 = separate exchange launches
                        dk["exchange_form"] = int(os.environ["GMPNP_BENCH_EXCHANGE_FORM"])
                    prun = PoreRun(partition=(world, rank), device_kwargs=dk, **common)
                except Exception as e:  # noqa: BLE001
                    err = "%s: %s" % (type(e).__name__, str(e)[:300])
                if agree(err is not None):
                    raise RuntimeError(err or "another rank could not set the transport up")
                state["phase"] = "%s: selftest" % transport
                try:
                    dev_err = prun.sys.ps.selftest()
                    err = None if dev_err == 0.0 else "selftest: largest deviation %.3e" % dev_err
                except Exception as e:  # noqa: BLE001
                    err = "selftest: %s: %s" % (type(e).__name__, str(e)[:300])
                check["selftest"] = "pass" if err is None else err
                if agree(err is not None):
                    raise RuntimeError(err or "selftest failed on another rank")
                state["phase"] = "%s: timed run" % transport
                pdt = timed(prun)
                pits, pkry = float(sum(prun.newton_its)), float(prun.sys.krylov_iterations)
                # every rank holds the same global history (all-gathered vertex values), so every rank decides alike
                if pits != float(sum(run.newton_its)):
                    raise RuntimeError("%d Newton iterations instead of the single-GPU run's %d" % (pits, sum(run.newton_its)))
                rel = float(np.linalg.norm(prun.history[-1] - final_replica) / np.linalg.norm(final_replica))
                check["state_vs_single_gpu"] = rel
                if not rel < 1e-8:
                    raise RuntimeError("final state differs from the single-GPU run's by %.3e (relative)" % rel)
                tt = torch.tensor([pdt], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                res = {"value": pits / float(tt[0]), "seconds": float(tt[0]), "newton_iterations": pits, "krylov_iterations": pkry,
                       "ms_per_step": 1e3 * float(tt[0]) / a.steps, "transport": transport, "state_vs_single_gpu": rel}
                if transport in ("peer", "peer-separate"):   # 2 = the exchange rides inside the next half-iteration's launch, 1 = separate launches
                    res["exchange_form"] = prun.sys.ps.exchange_form()
                check["run"] = "pass"
                tried.append({"transport": transport, "value": res["value"], "krylov_iterations": pkry})
                if state["best"] is None or res["value"] > state["best"]["value"]:   # identical on every rank (all-reduced time)
                    state["best"] = res
            except Exception as e:  # noqa: BLE001
                msg = "%s transport: %s: %s" % (transport, type(e).__name__, str(e)[:300])
                errors.append(msg)
                if check["run"] is None:
                    check["run"] = msg
                # every rank must take the same branch: an error on one rank only would leave the others in a collective,
                # where the watchdog ends the phase (exit code 3)
            finally:
                if prun is not None:
                    try:
                        prun.sys.close()
                    except Exception:  # noqa: BLE001
                        pass
        state["phase"] = "done"
        part = dict(state["best"]) if state["best"] is not None else {"error": "; ".join(errors) or "no transport"}
        part["transport_checks"] = state["checks"]
        if "value" in part:
            part["transports_timed"] = tried
            if errors:
                part["earlier_errors"] = errors
        wd.cancel()

    emit(part)
    run.sys.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def make_output(a, run, dev, prof, nv, world, dt, its, kry):
    """Everything of the JSON line that does not depend on the partitioned phase."""
    nb, nd = dev.n_blocks, dev.ndof
    nf = dev.nf
    alg_bytes = (nf * nf * 8) * nb + 4 * nb + 4 * (nv + 1) + 16 * nd  # SURVEY §8d, one SpMV
    mean_us = prof["mean_us"] if prof["sampled"] else dev.time_kernel(4, 200)
    launches = dev.krylov_launches_per_iteration
    ml_note = None
    if a.multilevel and a.refine > 0:
        # the live event brackets span whole half-iterations, i.e. the ~35 small launches of the multilevel term as well; the
        # roofline line is about the streaming kernel, so it is timed by itself here (back-to-back launches, HIP events)
        ml_note = "live half-iteration incl. the multilevel launches: %.1f us; the tile kernels alone (k_bicg_a_mat / k_bicg_b_mat, back-to-back): see mean_launch_us" % mean_us
        mean_us = 0.5 * (dev.time_kernel(14, 50) + dev.time_kernel(15, 50))
    if launches == 2:
        kernel_name = ("k_half_a / k_half_b (one launch per BiCGStab half-iteration: the coarse workgroups ride in front of "
                       "the tile workgroups = SELL node-block SpMV + vector updates, fp64; the launch duration includes "
                       "the in-launch wait for the coarse result, which the 4-launch form spends in a separate launch)")
    else:
        kernel_name = "k_bicg_a / k_bicg_b (fused BiCGStab half-iteration = SELL node-block SpMV + vector updates, fp64)"
    if a.multilevel and a.refine > 0:
        kernel_name = "k_bicg_a_mat / k_bicg_b_mat (tile kernels of the materialised vector form: SELL node-block SpMV staging one vector, fp64)"
    achieved = alg_bytes / (mean_us * 1e-6) / 1e9
    # what an EMPTY start/stop event pair measures on this stream: each timed burst carries that much ONCE (a burst is 2 x
    # ~20 launches), reported for information
    try:
        ev_overhead = dev.event_overhead(200)
    except Exception:  # noqa: BLE001
        ev_overhead = None
    # memory-side bytes per launch from the committed PMC passes; only valid for the build they were measured on
    traffic, traffic_note = None, "no PMC file"
    pmc = os.path.join(ROOT, "profiles", "spmv_pmc.json")
    if os.path.exists(pmc) and a.mesh == "L_50_R_5" and a.refine == 0:
        with open(pmc) as fh:
            pj = json.load(fh)
        if pj.get("build_id") == dev.build_id and pj.get("launches_per_krylov_iteration") == launches:
            traffic, traffic_note = pj.get("hbm_bytes_per_launch"), pj.get("source")
        else:
            traffic_note = "profiles/spmv_pmc.json was measured on build %s, this library is %s: dropped" % (pj.get("build_id"), dev.build_id)
    out = {
        "metric": "newton_iterations_per_sec", "value": its / dt, "unit": "Newton-iterations/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "reference inputs shipped in data/utilities (%s mesh, parameters_pore.yaml, bulk_soln_0.5KHCO3.yaml); "
                "deterministic, no RNG" % a.mesh,
        "config": {"refine": a.refine, "n_vertices": nv,
                   "preconditioner": ("node-block Jacobi + 8 slab aggregates + geometric multilevel term over %d nested meshes (V(1,1) cycles on the coarser levels)" % (a.refine + 1))
                   if (a.multilevel and a.refine > 0) else "node-block Jacobi + 8 slab aggregates (two-level)",
                   "workload": "3D MPNP_CO2ER_pore %s, 0.5 M KHCO3, K+, V=-1: time steps 0..%d from t=0 "
                               "(Newton rtol=atol=1e-4, omega=0.9, max 50; linear solve = two-level BiCGStab to "
                               "1e-10 relative residual)" % (a.mesh, a.steps - 1),
                   "n_dofs": nd, "jacobian_nnz": dev.jacobian_nnz, "newton_iterations": its,
                   "krylov_iterations": kry,
                   "parallelism": "1 GPU" if world == 1 else "%d independent replicas, one per GPU (no collective)" % world},
        "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved,
                     "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_note,
                     "algorithmic_bytes_per_launch": alg_bytes, "mean_launch_us": mean_us,
                     "empty_event_pair_us": ev_overhead,   # information only: `achieved` uses the raw event time (conservative)
                     "timing": "one HIP event pair around the first burst of every Nth solve (back-to-back live half-iterations); mean = elapsed / half-iterations, launch gaps included",
                     "launches_sampled": prof["sampled"], "launches_total": prof["launched"],
                     "launches_per_krylov_iteration": launches, "multilevel_note": ml_note},
    }
    if world == 1:
        out["scaling_note"] = "N = 1: one problem on one GPU; the field says weak because the contract has two values (at N > 1: strong = ONE problem partitioned, replicas = weak)"
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(run)
    if world > 1:
        out["replicas"] = {"metric": "newton_iterations_per_sec, %d independent L_50_R_5 problems, one per GPU (BASELINE configs[4] mapping)" % world,
                           "value": its / dt, "ms_per_step": 1e3 * dt / a.steps, "newton_iterations": its, "krylov_iterations": kry, "scaling": "weak"}
    return out


def attach_partitioned(out, a, world, part):
    """The line with the result of the partitioned phase attached (pure dictionary work: also run by the watchdog)."""
    out = dict(out)
    if world == 1:
        if part:
            out["partitioned_rehearsal"] = part
        return out
    if part and "value" in part:
        # the headline at N > 1: ONE problem on N GPUs (strong scaling; Newton iterations counted once)
        out.update(value=part["value"], ms_per_step=part["ms_per_step"], scaling="strong")
        out["config"] = dict(out["config"], newton_iterations=part["newton_iterations"], krylov_iterations=part["krylov_iterations"],
                             parallelism="ONE problem, %d z-slab mesh partitions, one per rank: ghost-row exchange + one fused all-reduce per "
                                         "BiCGStab half-iteration (%s), global coarse space (gmpnp_group_newton_solve)"
                                         % (world, {"rccl": "RCCL on the solver's stream",
                                                    "peer": "peer mailboxes: stores into the other ranks' IPC-mapped memory over xGMI; the exchange of a half-iteration rides inside the next launch as flagged words",
                                                    "peer-separate": "peer mailboxes: stores into the other ranks' IPC-mapped memory over xGMI, one exchange launch per half-iteration"}.get(part.get("transport"), "host-staged transport over torch.distributed")))
        out["roofline"] = dict(out["roofline"], note="kernel durations sampled in the replica phase (same kernels, whole mesh per GPU)")
        out["partitioned"] = {k: part[k] for k in ("transport", "exchange_form", "seconds", "transports_timed", "earlier_errors", "note", "transport_checks",
                                                   "state_vs_single_gpu", "watchdog") if k in part}
    else:
        out["partitioned"] = part or {"error": "not run (--replicas-only)"}
    return out


if __name__ == "__main__":
    main()
