"""The published integrands, evaluated point by point, against the coefficient-table form the oracle and the HIP kernels
both consume.

Everything else in the test suite compares the GPU with `oracle/gmpnp_oracle.py`, and both are fed by the SAME tables
(`gmpnp_amd.params` -> `Model`: rc0/rc1/rc2, z, a, qzb, epsc, wall_flux, exit_kappa, point_flux, inv_dt).  A wrong sign
or a wrong scale in that shared layer would be invisible to all of them.  This file does not touch those tables.  It

 1. recomputes the scalars of the reference's parameter block from the YAML inputs with NAMED quantities, following
    3D/MPNP_CO2ER_pore.py:126-324,358-365,470-499 and 1D/MPNP_CO2ER_EDL.py:89-208,256-268,366-375 line by line
    (`PoreConstants`, `EdlConstants` below; plain floats and dicts keyed by species name, no `Model`);
 2. evaluates the weak-form INTEGRANDS of 3D:505-767 (incl. the ds(2)/ds(3) lines 560,588,616,644,671,698,724,750) and
    1D:383-595 (+ the J_OH, J_H terms added at the solve, 1D:738) at quadrature points as written there: rates R_X from
    named rate constants, the steric quotient u_X / (1 - sum_j a_j u_j) times dot(sum_j a_j grad u_j, grad v), the
    permittivity expression with its literal 55 / 6 / 1e-3 constants, the charge sum, the Robin exit fluxes;
 3. integrates them with the rule the form compiler would use (degree 3 for F; for the Jacobian the derivative of the
    integrand with the degree-4 rule) and compares with `oracle.element_residual_jacobian` /
    `oracle.facet_terms` called with the product's `Model` on random states: 1e-12 (F), 1e-7 (J by central differences).

The quadrature POINTS are still the shared tables of `gmpnp_amd.model.default_quadrature` (FIAT's schemes restated from
memory: that ingredient stays "parity unpinned"); all polynomial terms are exact under any rule of
the degree, so only the steric quotient depends on them.
"""
import os

import numpy as np
import pytest
import yaml

import gmpnp_oracle as O
from conftest import ROOT
from gmpnp_amd.model import default_quadrature
from gmpnp_amd.params import edl_parameters, pore_parameters

UTIL = os.path.join(ROOT, "data", "utilities")


def _load(name):
    with open(os.path.join(UTIL, name)) as fh:
        return yaml.safe_load(fh)


# ---------------------------------------------------------------------------------------------------------------
# parameter blocks, recomputed with the reference's names
# ---------------------------------------------------------------------------------------------------------------
class PoreConstants:
    """3D/MPNP_CO2ER_pore.py:126-324 (scaling), :358-365 (time step), :470-499 (fluxes)."""

    def __init__(self, concentration_elec=0.5, L=50e-9, R=5e-9, cation="K", voltage_multiplier=-1.0, H2_FE=0.05,
                 current_rough=3000.0, porosity_eff=0.5, tortuosity_eff=1.5, constrictivity_eff=0.9, press_gas=1.0,
                 pore_geom_multiplier=1.0, electrolyte_flow_geom_multiplier=1.0, y_CO2=0.95, roughness_factor=150.0):
        d = _load("parameters_pore.yaml")
        self.cat = cat = cation
        self.species = sp = ["H", "OH", "HCO3", "CO32", "CO2", "CO", "H2", cat]
        self.k = dict(d["rate_constants"])                                                     # :126-134
        D = {s: d["diff_coef"]["D_" + s] for s in sp}                                          # :147-148
        Deff = {s: D[s] * porosity_eff * constrictivity_eff * pore_geom_multiplier / tortuosity_eff ** 2 for s in sp}  # :156-158
        self.n_water = {"H": d["Hydration_number"]["w_H"], cat: d["Hydration_number"]["w_" + cat]}  # :161-164
        size = {s: d["solv_size"]["a_" + s] for s in sp}                                       # :170-171
        nc, sy = d["nat_const"], d["sys_params"]
        farad, eps_0, R_gas, N_A = nc["F"], nc["eps_0"], nc["R"], nc["N_A"]
        self.eps_rel = nc["eps_rel"]
        temp, density_e, viscosity_e = sy["T"], sy["density_e"], sy["viscosity_e"]
        L_electrode, vel_e, A_cross_e, L_cross_e = sy["L_electrode"], sy["vel_e"], sy["A_cross_e"], sy["L_cross_e"]
        y_CO = 0.9 * (1 - y_CO2)                                                               # :218
        y_H2 = 1 - y_CO2 - y_CO
        b = _load("bulk_soln_%sKHCO3.yaml" % str(concentration_elec))["bulk_conc_pre_CO2"]["concentrations"]  # :224-238
        bulk = {s: b["C0_" + s] for s in sp}
        eq_CO2 = d["Henrys_const"]["H_CO2"] * press_gas * y_CO2 * density_e                    # :253-255
        eq_CO = d["Henrys_const"]["H_CO"] * press_gas * y_CO * density_e
        eq_H2 = d["Henrys_const"]["H_H2"] * press_gas * y_H2 * density_e
        bulk["CO"], bulk["H2"] = 0.01 * eq_CO, 0.01 * eq_H2                                    # :258-259
        self.bulk = bulk
        self.eq_scaled = {"CO2": eq_CO2 / bulk["CO2"], "CO": eq_CO / bulk["CO"], "H2": eq_H2 / bulk["H2"]}  # :261-263
        self.z = {"H": 1, "OH": -1, "HCO3": -1, "CO32": -2, "CO2": 0, "CO": 0, "H2": 0, cat: 1}  # :233-234
        time_constant = L ** 2 / Deff["CO32"]                                                  # :270
        self.scale_R = {s: L ** 2 / (Deff[s] * bulk[s]) for s in sp}                           # :276-277
        self.q = farad ** 2 * L ** 2 / (eps_0 * R_gas * temp)                                  # :280
        self.scale_vol = {s: size[s] ** 3 * bulk[s] * N_A for s in sp}                         # :286-287
        J_pref = {s: L / (Deff[s] * bulk[s]) for s in sp}                                      # :293-295
        Re = density_e * (vel_e / A_cross_e) * L_electrode * electrolyte_flow_geom_multiplier / viscosity_e  # :301-302
        k_elec = {}
        for s in sp:                                                                           # :317-321
            Sc = viscosity_e / (density_e * D[s])
            Sh = 1.017 * ((L_electrode * 2 / L_cross_e) * Re * Sc) ** (1.0 / 3)
            k_elec[s] = (D[s] / L_electrode) * Sh
        self.del_t = 1.0e-3 / time_constant                                                    # :358-365
        CO_FE = 1 - H2_FE                                                                      # :469
        current_planar = current_rough / roughness_factor
        self.J_wall = {"CO2": (J_pref["CO2"] / farad) * current_planar * 0.5 * CO_FE,          # :474-481
                       "CO": (J_pref["CO"] / farad) * current_planar * 0.5 * CO_FE * (-1.0),
                       "H2": (J_pref["H2"] / farad) * current_planar * 0.5 * H2_FE * (-1.0),
                       "OH": (J_pref["OH"] / farad) * current_planar * (-1.0)}
        self.kappa_exit = {s: J_pref[s] * k_elec[s] * bulk[s] for s in sp}                     # :484-499
        self.voltage_scaled = voltage_multiplier


class EdlConstants:
    """1D/MPNP_CO2ER_EDL.py:89-208 (scaling), :256-268 (dry-run time step), :366-375 (OHP fluxes)."""

    def __init__(self, concentration_elec=0.1, voltage_multiplier=-1.0, H2_FE=0.2, current_OHP_ss=10.0, L_n=50e-6,
                 cation="K", current_H_frac=0.0):
        d = _load("parameters.yaml")
        self.cat = cat = cation
        self.species = sp = ["H", "OH", "HCO3", "CO32", "CO2", cat]
        self.k = dict(d["rate_constants"])                                                     # :92-100
        self.n_water = {"H": 10.0, cat: {"K": 4, "Li": 5, "Cs": 3, "Na": 5}[cat]}              # :106-115
        D = {s: d["diff_coef"]["D_" + s] for s in sp}                                          # :123-124
        size = {s: d["solv_size"]["a_" + s] for s in sp}                                       # :130-131
        nc = d["nat_const"]
        farad, temp, k_B, e_0, eps_0, R, N_A = nc["F"], nc["T"], nc["k_B"], nc["e_0"], nc["eps_0"], nc["R"], nc["N_A"]
        self.eps_rel = nc["eps_rel"]
        b = _load("bulk_soln_%sKHCO3.yaml" % str(concentration_elec))["bulk_conc_post_CO2"]["concentrations"]  # :146-160
        self.bulk = bulk = {s: b["C0_" + s] for s in sp}
        self.z = {"H": 1, "OH": -1, "HCO3": -1, "CO32": -2, "CO2": 0, cat: 1}                  # :157
        L_debye = np.sqrt(eps_0 * self.eps_rel * k_B * temp / (2 * e_0 ** 2 * concentration_elec * 1.0e+3 * N_A))  # :173-176
        self.L_D = L_debye / L_n                                                               # :178
        time_constant = L_debye * L_n / D["CO32"]                                              # :183
        self.scale_R = {s: L_n ** 2 / (D[s] * bulk[s]) for s in sp}                            # :189-190
        self.q = farad ** 2 * L_n ** 2 / (eps_0 * R * temp)                                    # :193
        self.scale_vol = {s: size[s] ** 3 * bulk[s] * N_A for s in sp}                         # :199-200
        J_H_pref = L_n / (D["H"] * bulk["H"] * farad)                                          # :203-205
        J_OH_pref = L_n / (D["OH"] * bulk["OH"] * farad)
        J_CO2_pref = L_n / (D["CO2"] * bulk["CO2"] * farad)
        self.del_t = 1.0e-5 / time_constant                                                    # :260-267 (dry run)
        CO_FE = 1 - H2_FE                                                                      # :368
        self.J_point = {"CO2": J_CO2_pref * current_OHP_ss * 0.5 * CO_FE,                      # :371-375
                        "OH": J_OH_pref * current_OHP_ss * (1 - current_H_frac) * (-1.0),
                        "H": J_H_pref * current_OHP_ss * current_H_frac}
        self.voltage_scaled = voltage_multiplier


# ---------------------------------------------------------------------------------------------------------------
# integrands
# ---------------------------------------------------------------------------------------------------------------
def production_rates(c, u):
    """R_X of 3D:505-532 / 1D:383-410 at one point; u: dict species -> scaled concentration."""
    k, B, s = c.k, c.bulk, c.scale_R
    H, OH, HCO3, CO32, CO2 = (u[x] * B[x] for x in ("H", "OH", "HCO3", "CO32", "CO2"))   # u_X * bulk_conc[X]
    R = {x: 0.0 for x in c.species}
    R["H"] = -s["H"] * (k["kw2"] * H * OH - k["kw1"])
    R["OH"] = -s["OH"] * (k["kw2"] * H * OH + k["ka1"] * OH * HCO3 + k["kb1"] * CO2 * OH - k["kw1"] - k["ka2"] * CO32 - k["kb2"] * HCO3)
    R["HCO3"] = -s["HCO3"] * (k["ka1"] * OH * HCO3 + k["kb2"] * HCO3 - k["ka2"] * CO32 - k["kb1"] * CO2 * OH)
    R["CO32"] = -s["CO32"] * (k["ka2"] * CO32 - k["ka1"] * OH * HCO3)
    R["CO2"] = -s["CO2"] * (k["kb1"] * CO2 * OH - k["kb2"] * HCO3)
    return R


def cell_integrand(c, u, gu, un, p_grad, v_species, v_val, v_grad, steric=True, time_scale=1.0):
    """Value at one point of the dx integrand of F_X (X = v_species) or of F_p (v_species == 'p') for the test function
    with value v_val and gradient v_grad.  u, un: dict species -> value; gu: dict species -> gradient (array);
    p_grad: gradient of the potential.  `time_scale` = 1 (3D: / del_t) or L_D (1D: / (del_t * L_D))."""
    if v_species == "p":                                                             # 3D:752-767, 1D:412-427
        cat = c.cat
        w = (c.n_water[cat] * u[cat] * c.bulk[cat] + c.n_water["H"] * u["H"] * c.bulk["H"]) * 1.0e-3
        eps = c.eps_rel * ((55 - w) / 55) + 6 * (w / 55)
        charge = sum(c.z[x] * u[x] * c.bulk[x] for x in ("H", "OH", "HCO3", "CO32", cat))
        return -eps * np.dot(p_grad, v_grad) + charge * c.q * v_val
    X = v_species
    val = ((u[X] - un[X]) / (c.del_t * time_scale)) * v_val                          # 3D:534, 1D:458
    val += np.dot(gu[X], v_grad)
    if c.z[X] != 0:                                                                   # the z = 0 species have no such line
        val += c.z[X] * u[X] * np.dot(p_grad, v_grad)
    val -= production_rates(c, u)[X] * v_val                                          # cation, CO, H2: no R term (R = 0 here)
    if steric:                                                                        # MPNP
        occupied = sum(c.scale_vol[x] * u[x] for x in c.species)
        crowd_grad = sum(c.scale_vol[x] * gu[x] for x in c.species)
        val += (u[X] / (1 - occupied)) * np.dot(crowd_grad, v_grad)
    return val


def _p1(X):
    """P1 basis on a simplex with vertex rows X: gradients (nn, d) and measure."""
    nn, d = X.shape
    A = np.concatenate([np.ones((nn, 1)), X], axis=1)
    G = np.linalg.inv(A)[1:, :].T          # row a = grad phi_a
    meas = abs(np.linalg.det(A)) / (1, 1, 2, 6)[d]
    return G, meas


def literal_cell_residual(c, X, U, Un, lam, w, steric=True, time_scale=1.0):
    """sum_q w_q |K| integrand(x_q) for every (node a, field): (nn, nf); fields = species..., potential."""
    G, meas = _p1(X)
    nn = X.shape[0]
    sp = c.species
    out = np.zeros((nn, len(sp) + 1))
    gu = {x: U[:, j] @ G for j, x in enumerate(sp)}
    gp = U[:, -1] @ G
    for lq, wq in zip(lam, w):
        u = {x: float(lq[:nn] @ U[:, j]) for j, x in enumerate(sp)}
        un = {x: float(lq[:nn] @ Un[:, j]) for j, x in enumerate(sp)}
        for a in range(nn):
            for j, x in enumerate(sp + ["p"]):
                out[a, j] += wq * meas * cell_integrand(c, u, gu, un, gp, x, lq[a], G[a], steric, time_scale)
    return out


# ---------------------------------------------------------------------------------------------------------------
# tests
# ---------------------------------------------------------------------------------------------------------------
def _random_cells(rng, nc, nn, d, ns, scale):
    X = rng.uniform(0, scale, (nc, nn, d))
    if d == 1:
        X = np.sort(X, axis=1)
    U = np.concatenate([rng.uniform(.5, 1.5, (nc, nn, ns)), rng.uniform(-1, 0, (nc, nn, 1))], 2)
    Un = np.concatenate([rng.uniform(.5, 1.5, (nc, nn, ns)), rng.uniform(-1, 0, (nc, nn, 1))], 2)
    return X, U, Un


PORE_CASES = [dict(concentration_elec=0.5, L=50e-9, R=5e-9),
              dict(concentration_elec=1.0, L=10e-9, R=5e-9, voltage_multiplier=-2.5, H2_FE=0.2, current_rough=1000.0),
              dict(concentration_elec=0.5, L=25e-9, R=5e-9, porosity_eff=0.4, tortuosity_eff=1.2, y_CO2=0.9, press_gas=2.0)]


@pytest.mark.parametrize("kw", PORE_CASES)
def test_pore_cell_residual_is_the_published_integrand(kw):
    """3D:505-767 at the degree-3 points vs the oracle's closed forms fed by gmpnp_amd.params tables."""
    c = PoreConstants(**kw)
    pp = pore_parameters(**kw)
    qd = default_quadrature(3)
    rng = np.random.default_rng(11)
    X, U, Un = _random_cells(rng, 5, 4, 3, 8, 0.05)
    Fe, _ = O.element_residual_jacobian(pp.model, qd, X, U, Un, want_jacobian=False)
    for e in range(X.shape[0]):
        lit = literal_cell_residual(c, X[e], U[e], Un[e], qd.lam_f, qd.w_f)
        assert np.allclose(lit, Fe[e], rtol=1e-11, atol=1e-11 * np.abs(Fe[e]).max()), (e, np.abs(lit - Fe[e]).max())


@pytest.mark.parametrize("kw", PORE_CASES[:2])
def test_pore_cell_jacobian_is_the_derivative_of_the_published_integrand(kw):
    """derivative(F, u) with the degree-4 rule (what FFC generates for the Jacobian form) by central differences of the
    literal integrand vs the oracle's analytic Jacobian."""
    c = PoreConstants(**kw)
    pp = pore_parameters(**kw)
    qd = default_quadrature(3)
    rng = np.random.default_rng(12)
    X, U, Un = _random_cells(rng, 2, 4, 3, 8, 0.05)
    _, Je = O.element_residual_jacobian(pp.model, qd, X, U, Un)
    for e in range(X.shape[0]):
        J = np.zeros((4, 9, 4, 9))
        for b in range(4):
            for j in range(9):
                h = 1e-6 * max(1.0, abs(U[e, b, j]))
                Up, Um = U[e].copy(), U[e].copy()
                Up[b, j] += h
                Um[b, j] -= h
                J[:, :, b, j] = (literal_cell_residual(c, X[e], Up, Un[e], qd.lam_j, qd.w_j)
                                 - literal_cell_residual(c, X[e], Um, Un[e], qd.lam_j, qd.w_j)) / (2 * h)
        scale = np.abs(Je[e]).max()
        assert np.abs(J - Je[e]).max() < 2e-7 * scale, np.abs(J - Je[e]).max() / scale


def test_pore_boundary_terms_are_the_published_flux_lines(pore10):
    """ds(2): J_X_wall v_X for OH, CO2, CO, H2 (3D:588,671,724,750); ds(3): J_pore_exit_X v_X = kappa_X (u_X - 1) v_X for
    all eight species (3D:560-750), integrated per facet with the 3-point edge-midpoint rule (exact for degree 2)."""
    pp, mesh, prob, bnd = pore10
    c = PoreConstants(concentration_elec=0.5, L=10e-9, R=5e-9)
    rng = np.random.default_rng(13)
    nv, nf = mesh.num_vertices, 9
    u2d = np.concatenate([rng.uniform(.5, 1.5, (nv, 8)), rng.uniform(-1, 0, (nv, 1))], 1)
    Fb, _ = O.facet_terms(prob, u2d, want_jacobian=False)
    lit = np.zeros(nv * nf)
    mid = np.array([[.5, .5, 0], [0, .5, .5], [.5, 0, .5]])
    sp = c.species

    def area(f):
        x = mesh.coords[f]
        return 0.5 * np.linalg.norm(np.cross(x[1] - x[0], x[2] - x[0]))

    for f in prob.wall_facets:
        ar = area(f)
        for x, Jw in c.J_wall.items():
            j = sp.index(x)
            for lq in mid:
                for a in range(3):
                    lit[f[a] * nf + j] += (ar / 3.0) * Jw * lq[a]
    for f in prob.exit_facets:
        ar = area(f)
        for j, x in enumerate(sp):
            for lq in mid:
                ux = lq @ u2d[f, j]
                for a in range(3):
                    lit[f[a] * nf + j] += (ar / 3.0) * c.kappa_exit[x] * (ux - 1) * lq[a]
    assert len(prob.wall_facets) > 100 and len(prob.exit_facets) > 100
    assert np.allclose(lit, Fb, rtol=1e-11, atol=1e-12 * np.abs(Fb).max())
    # and the Dirichlet values of bc3..bc6 (3D:460-465)
    assert np.isclose(pp.voltage_scaled, c.voltage_scaled)
    assert np.isclose(pp.eq_conc_CO2_scaled, c.eq_scaled["CO2"], rtol=1e-13)
    assert np.isclose(pp.eq_conc_CO_scaled, c.eq_scaled["CO"], rtol=1e-13) and np.isclose(pp.eq_conc_H2_scaled, c.eq_scaled["H2"], rtol=1e-13)


EDL_CASES = [dict(), dict(L_n=1e-6, cation="Cs", voltage_multiplier=-10.0),
             dict(L_n=10e-6, cation="Li", H2_FE=0.4, current_OHP_ss=50.0), dict(L_n=5e-6, cation="Na", model="PNP")]


@pytest.mark.parametrize("kw", EDL_CASES)
def test_edl_cell_residual_and_point_fluxes_are_the_published_integrand(kw):
    """1D:383-595 (MPNP, and the PNP variant 1D:429-455) at the 2-point Gauss rule; OHP point terms 1D:371-375,553,738."""
    kw = dict(kw)
    model = kw.pop("model", "MPNP")
    c = EdlConstants(**kw)
    ep = edl_parameters(model=model, **kw)
    qd = default_quadrature(1)
    rng = np.random.default_rng(14)
    X, U, Un = _random_cells(rng, 6, 2, 1, 6, 1e-3)
    Fe, _ = O.element_residual_jacobian(ep.model, qd, X, U, Un, want_jacobian=False)
    for e in range(X.shape[0]):
        lit = literal_cell_residual(c, X[e], U[e], Un[e], qd.lam_f, qd.w_f, steric=(model == "MPNP"), time_scale=c.L_D)
        assert np.allclose(lit, Fe[e], rtol=1e-11, atol=1e-11 * np.abs(Fe[e]).max()), (e, np.abs(lit - Fe[e]).max())
    # point fluxes: + J_CO2 v_CO2 ds (1D:553), + J_OH v_OH ds + J_H v_H ds (1D:738); ds at x = 0 only matters (Q6)
    pf = dict(zip(c.species, ep.model.point_flux))
    for x in c.species:
        assert np.isclose(pf[x], c.J_point.get(x, 0.0), rtol=1e-13, atol=0.0), x


def test_edl_cell_jacobian_is_the_derivative_of_the_published_integrand():
    c = EdlConstants(L_n=1e-6, cation="Cs", voltage_multiplier=-10.0)
    ep = edl_parameters(L_n=1e-6, cation="Cs", voltage_multiplier=-10.0)
    qd = default_quadrature(1)
    rng = np.random.default_rng(15)
    X, U, Un = _random_cells(rng, 3, 2, 1, 6, 1e-3)
    _, Je = O.element_residual_jacobian(ep.model, qd, X, U, Un)
    for e in range(X.shape[0]):
        J = np.zeros((2, 7, 2, 7))
        for b in range(2):
            for j in range(7):
                h = 1e-6 * max(1.0, abs(U[e, b, j]))
                Up, Um = U[e].copy(), U[e].copy()
                Up[b, j] += h
                Um[b, j] -= h
                J[:, :, b, j] = (literal_cell_residual(c, X[e], Up, Un[e], qd.lam_j, qd.w_j, time_scale=c.L_D)
                                 - literal_cell_residual(c, X[e], Um, Un[e], qd.lam_j, qd.w_j, time_scale=c.L_D)) / (2 * h)
        scale = np.abs(Je[e]).max()
        assert np.abs(J - Je[e]).max() < 2e-7 * scale, np.abs(J - Je[e]).max() / scale


def test_controller_updates_the_point_fluxes_like_the_reference():
    """1D:789-793: J_OH, J_H re-evaluated with the new current_H_frac."""
    ep = edl_parameters(L_n=10e-6, voltage_multiplier=-2.5, H_OHP=0.5, H2_FE=0.4)
    for frac in (0.001, 0.37, 1.0):
        c = EdlConstants(L_n=10e-6, voltage_multiplier=-2.5, H2_FE=0.4, current_H_frac=frac)
        JH, JOH = ep.ohp_fluxes(frac)
        assert np.isclose(JH, c.J_point["H"], rtol=1e-13) and np.isclose(JOH, c.J_point["OH"], rtol=1e-13)


def test_sechenov_feedback_is_the_published_formula():
    """The per-step host feedback of the 3D loop (3D:70-93 CO2_conc, 3D:817-835): Henry's constant ln K_H = 93.4517 (100/T)
    - 60.2409 + 23.3585 ln(T/100), h_CO2 = h0 + hT (T - 298.15), s = sum_ion (h_ion + h_CO2) c_ion / 1000 over OH, HCO3, CO32,
    cation with c_ion = median(u_ion) bulk_ion, C = f_CO2 K_H 1000 10^-s; the new bc4 value is C / bulk_CO2.  Recomputed here
    from the YAML numbers, against PoreParameters.sechenov_co2_scaled (what PoreRun.step and the oracle's time loop call)."""
    import math
    for kw in PORE_CASES:
        c = PoreConstants(**kw)
        pp = pore_parameters(**kw)
        d = _load("parameters_pore.yaml")
        T = d["sys_params"]["T"]
        hs = d["sechonov_const"]
        y_CO2, press = kw.get("y_CO2", 0.95), kw.get("press_gas", 1.0)
        lnK = 93.4517 * (100 / T) - 60.2409 + 23.3585 * math.log(T / 100)
        h_CO2 = hs["h_CO2_0"] + hs["h_CO2_T"] * (T - 298.15)
        rng = np.random.default_rng(3)
        for _ in range(5):
            med = dict(zip(("OH", "HCO3", "CO32", c.cat), rng.uniform(0.2, 3.0, 4)))
            s = sum((hs["h_ion_" + ion] + h_CO2) * (med[ion] * c.bulk[ion] / 1000) for ion in med)
            want = (y_CO2 * press) * math.exp(lnK) * 1000 * 10 ** (-s) / c.bulk["CO2"]
            got = pp.sechenov_co2_scaled(med["OH"], med["HCO3"], med["CO32"], med[c.cat])
            assert abs(got / want - 1.0) < 1e-13
