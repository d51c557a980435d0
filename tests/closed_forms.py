"""Closed-form cases for the 3D forms (3D/MPNP_CO2ER_pore.py:474-767), independent of who solves them: each function returns the
modified problem, the start state and a `check(state)` that compares a steady / one-step solution with the closed form.
tests/test_gpu_parity.py solves them with the HIP library, tests/test_oracle_pins.py with the CPU oracle.

  boltzmann_case   transport: zero-flux equilibrium u_i = (1 - S) / (1 - S_b) exp(-z_i p) at every vertex, whatever p
  bessel_case      Poisson coupling: Debye-Hueckel profile p = V I0(kappa r) / I0(kappa R) in the long pore
  rates_case       reactions + time term: a uniform neutral state follows (u - u^n) / del_t = R(u), R literal (test_literal_forms)
  flux_case        wall Neumann / exit Robin terms: exact discrete flux balance with the literal J_X_wall, kappa_X
"""
import copy

import numpy as np

from gmpnp_amd.mesh import read_dolfin_xml, resolve_mesh_path
from gmpnp_amd.params import pore_parameters, utilities_dir
from gmpnp_amd.problem import pore_problem


def sp_tight(rtol, atol):
    return {"nonlinear_solver": "newton", "newton_solver": {"linear_solver": "mumps", "maximum_iterations": 50, "relative_tolerance": rtol,
                                                            "absolute_tolerance": atol, "relaxation_parameter": 1.0}}


def _base(L, R, refine, reactions=False, wall_flux=False, exit_flux=True, q_scale=0.01, steady=True, coarse=None):
    """`coarse` = (n_rings, n_layers): a small generated cylinder (gmpnp_amd.meshgen) instead of the reference mesh, marked with a
    wall tolerance wide enough for its chords — the size the CPU oracle solves in seconds."""
    pp = pore_parameters(concentration_elec=0.5, L=L, R=R)
    if coarse is None:
        mesh = read_dolfin_xml(resolve_mesh_path(utilities_dir(), pp.mesh_name))
        prob, _ = pore_problem(pp, mesh, refine=refine)
    else:
        from gmpnp_amd.mesh import mark_pore_boundaries
        from gmpnp_amd.meshgen import cylinder_mesh
        from gmpnp_amd.problem import Problem, pore_dirichlet
        mesh = cylinder_mesh(pp.aspect_pore, coarse[0], coarse[1])
        sag = pp.aspect_pore ** 2 * (1.0 - np.cos(np.pi / (6 * coarse[0])) ** 2)      # r^2 deficit of a rim chord's midpoint
        bnd = mark_pore_boundaries(mesh, pp.aspect_pore, 1.5 * sag)
        assert bnd.counts[2][0] == 2 * 6 * coarse[0] * coarse[1] and bnd.counts[2][1] == 0      # every lateral facet, no interior one
        dofs, vals = pore_dirichlet(pp, bnd)
        prob = Problem(coords=mesh.coords, cells=mesh.cells, model=pp.model, wall_facets=bnd.ds_facets[2], exit_facets=bnd.ds_facets[3],
                       bc_dofs=dofs, bc_vals=vals)
    m = copy.deepcopy(prob.model)
    if not reactions:
        m.rc0[:] = 0.0; m.rc1[:] = 0.0; m.rc2[:] = 0.0
    if not wall_flux:
        m.wall_flux = np.zeros_like(m.wall_flux)
    if not exit_flux:
        m.exit_kappa = np.zeros_like(m.exit_kappa)
    if steady:
        m.inv_dt = 1e-9
    m.q = m.q * q_scale
    prob = copy.copy(prob)
    prob.model = m
    ns = m.n_species
    keep = (prob.bc_dofs % (ns + 1)) == ns                     # potential conditions only: no species Dirichlet condition
    wall = prob.bc_vals[keep] != 0.0                           # S2 carries the applied voltage, S1 and S3 carry 0
    prob.bc_dofs, prob.bc_vals = prob.bc_dofs[keep], prob.bc_vals[keep]
    return prob, m, ns, prob.coords.shape[0], wall


def boltzmann_case(refine=0, V=-1.0, coarse=None):
    """Reactions and wall fluxes off, Robin exit towards 1 on S3 (where p = 0), wall potential V sin^2(pi z) (no jump against
    p = 0 on S1 / S3), Debye length x 10 (q / 100: the relation does not contain q), dt -> infinity."""
    prob, m, ns, nv, wall = _base(10e-9, 5e-9, refine, coarse=coarse)
    zc = prob.coords[prob.bc_dofs // (ns + 1), 2]
    prob.bc_vals = np.where(wall, V * np.sin(np.pi * zc) ** 2, 0.0)
    z, a = np.asarray(m.z), np.asarray(m.a)

    def check(state):
        u = np.asarray(state).reshape(nv, ns + 1)
        p, c = u[:, ns], u[:, :ns]
        assert abs(p.min() - V) < 1e-3 and abs(p.max()) < 1e-6
        S = c @ a
        expect = ((1.0 - S) / (1.0 - a.sum()))[:, None] * np.exp(-z[None, :] * p[:, None])
        assert c[:, z > 0].max() > np.exp(-0.8 * V) and c[:, z < -1.5].min() < np.exp(1.6 * V)   # a real double layer: cations piled up, CO3-- driven out
        d = c - expect
        return np.abs(d).max(), np.sqrt((d ** 2).mean()), np.abs(c[:, z == 0] / expect[:, z == 0] - 1.0).max()

    return prob, np.tile(np.r_[np.ones(ns), 0.0], nv), check


def bessel_case(refine=0, V=-0.005, coarse=None):
    """Small uniform wall potential in the long L_50_R_5 pore (aspect 10: the ends are five diameters from the middle), q / 100 so
    that kappa R = O(1) is resolved: eps_b lap(p) = q (sum_i z_i^2 bulk_i) p, kappa^2 = q sum_i z_i^2 bulk_i / eps_b."""
    from scipy.special import i0
    prob, m, ns, nv, wall = _base(50e-9, 5e-9, refine, coarse=coarse)
    prob.bc_vals = np.where(wall, V, 0.0)
    z, bulk = np.asarray(m.z), np.asarray(m.bulk)
    eps_b = m.eps0 + float(np.sum(m.epsc))
    kappa = np.sqrt(m.q * float(np.sum(z * z * bulk)) / eps_b)
    r = np.hypot(prob.coords[:, 0], prob.coords[:, 1])
    R = 0.1                                                        # R / L
    mid = (prob.coords[:, 2] > 0.3) & (prob.coords[:, 2] < 0.7) & (r < 0.999 * R)
    assert 0.5 < kappa * R < 5.0 and mid.sum() > (200 if coarse is None else 50)

    def check(state):
        u = np.asarray(state).reshape(nv, ns + 1)
        d = u[mid, ns] / V - i0(kappa * r[mid]) / i0(kappa * R)
        return np.abs(d).max(), np.sqrt((d ** 2).mean()), float((u[mid, ns] / V).min())

    return prob, np.tile(np.r_[np.ones(ns), 0.0], nv), check


def rates_case(coarse=None):
    """Uniform state, p = 0 on the whole boundary, no wall or exit flux, the product's own time step: start from the bulk with
    twice the protons (and the hydroxide that keeps it electroneutral) and 50 % more dissolved CO2; one backward-Euler step."""
    from scipy.optimize import fsolve
    from test_literal_forms import PoreConstants, production_rates
    prob, m, ns, nv, wall = _base(10e-9, 5e-9, 0, reactions=True, exit_flux=False, q_scale=1.0, steady=False, coarse=coarse)
    prob.bc_vals = np.zeros(len(prob.bc_dofs))
    c = PoreConstants(concentration_elec=0.5, L=10e-9, R=5e-9)
    assert list(m.species) == c.species
    iH, iOH, iCO2 = (c.species.index(x) for x in ("H", "OH", "CO2"))
    un = np.ones(ns)
    un[iCO2] = 1.5
    un[iH] = 2.0
    un[iOH] = 1.0 + c.bulk["H"] / c.bulk["OH"]

    def step(u):
        R = production_rates(c, dict(zip(c.species, u)))
        return (u - un) / c.del_t - np.array([R[x] for x in c.species])

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        u1 = fsolve(step, un, xtol=1e-13)
    for _ in range(3):                                                   # polish: Newton with a finite-difference Jacobian
        J = np.empty((ns, ns))
        for j in range(ns):
            e = np.zeros(ns); e[j] = 1e-7 * max(1.0, abs(u1[j]))
            J[:, j] = (step(u1 + e) - step(u1 - e)) / (2 * e[j])
        u1 = u1 - np.linalg.solve(J, step(u1))
    assert np.abs(step(u1)).max() < 1e-9 * np.abs((u1 - un) / c.del_t).max()
    moved = np.abs(u1 - un) / un
    assert moved[iH] > 0.05 and moved[iCO2] > 1e-9          # the protons recombine within the step; CO2 hydration barely starts
    neutral = np.asarray(m.z) == 0

    def check(state):
        u = np.asarray(state).reshape(nv, ns + 1)
        dev = np.abs(u[:, :ns] / u1[None, :] - 1.0).max(0)
        return dev.max(), dev[neutral].max(), np.abs(u[:, ns]).max()

    return prob, np.tile(np.r_[un, 0.0], nv), check


def flux_case(coarse=None):
    """Wall sources for CO and H2 only, Robin exit for all, gases without their Dirichlet condition, reduced wall voltage, dt ->
    infinity.  For X in (CO, H2): J_X_wall |S2| + kappa_X int_S3 (u_X - 1) ds + (1 / dt) int (u_X - 1) dx = 0 exactly (sum of all
    test functions), J_X_wall and kappa_X literal (test_literal_forms.PoreConstants), areas and volumes from the mesh."""
    from test_literal_forms import PoreConstants
    prob, m, ns, nv, wall = _base(10e-9, 5e-9, 0, wall_flux=True, coarse=coarse)
    for X in ("OH", "CO2"):
        m.wall_flux[list(m.species).index(X)] = 0.0
    prob.bc_vals = 0.2 * prob.bc_vals
    c = PoreConstants(concentration_elec=0.5, L=10e-9, R=5e-9)

    def area(f):
        X = prob.coords[f]
        return 0.5 * np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1)

    a2, a3 = area(prob.wall_facets), area(prob.exit_facets)
    assert abs(a2.sum() / (2 * np.pi * 0.5) - 1.0) < (0.02 if coarse is None else 0.05)   # the wall of the R / L = 0.5 cylinder (3D/mesh_tests.py:80-85)
    Xc = prob.coords[prob.cells]
    vol = np.abs(np.linalg.det(Xc[:, 1:] - Xc[:, :1])) / 6.0

    def check(state):
        u = np.asarray(state).reshape(nv, ns + 1)
        out = {}
        for X in ("CO", "H2"):
            i = c.species.index(X)
            excess = float((a3 * (u[prob.exit_facets, i].mean(axis=1) - 1.0)).sum())      # P1: facet mean = mean of its vertex values
            stored = m.inv_dt * float((vol * (u[prob.cells, i].mean(axis=1) - 1.0)).sum())
            out[X] = (c.J_wall[X] * a2.sum() + c.kappa_exit[X] * excess + stored, abs(c.J_wall[X] * a2.sum()), excess / a3.sum())
        return out

    return prob, np.tile(np.r_[np.ones(ns), 0.0], nv), check
