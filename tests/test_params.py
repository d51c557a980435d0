"""L4 scalar tables against the values recorded in SURVEY §8c (computed there from the reference's formulas
3D/MPNP_CO2ER_pore.py:253-324 and 1D/MPNP_CO2ER_EDL.py:173-208)."""
import math

import numpy as np
import pytest

from gmpnp_amd.params import co2_conc, edl_parameters, pore_parameters


def rel(a, b):
    return abs(a - b) / abs(b)


def test_pore_scalars_survey_values():
    pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
    s = pp.scalars
    assert rel(s["q"], 1060.89) < 1e-5
    assert rel(s["time_constant"], 1.35428e-5) < 1e-5
    assert rel(s["dt"], 73.84) < 1e-5
    assert rel(pp.eq_conc_CO2_scaled, 5.745084) < 1e-6
    assert rel(pp.eq_conc_CO_scaled, 100.0) < 1e-12 and rel(pp.eq_conc_H2_scaled, 100.0) < 1e-12
    assert rel(s["J_wall"]["OH"], -4.807206) < 1e-6
    assert rel(s["J_wall"]["CO"], -28.44962) < 1e-6
    assert rel(s["J_wall"]["H2"], -7.404196) < 1e-6
    assert rel(s["J_wall"]["CO2"], 2.29916e-3) < 1e-5
    assert rel(s["scale_vol_sum"], 0.239754) < 1e-5
    assert pp.mesh_name == "L_50_R_5.xml" and pp.tot_num_steps == 1000
    assert pp.species == ["H", "OH", "HCO3", "CO32", "CO2", "CO", "H2", "K"]
    m = pp.model
    assert m.n_fields == 9 and m.dim == 3 and m.steric
    assert list(m.z) == [1, -1, -1, -2, 0, 0, 0, 1]
    assert rel(m.inv_dt, 1 / 73.84) < 1e-5
    # permittivity at bulk: eps_rel*(55-w)/55 + 6 w/55 with w = (4*500 + 10*4.89e-6)*1e-3
    w = (4 * 500.0 + 10 * pp.bulk_conc["H"]) * 1e-3
    assert rel(m.eps0 + m.epsc.sum(), 80.1 * (55 - w) / 55 + 6 * w / 55) < 1e-14
    # bulk state is in reaction equilibrium up to the YAML's rounding: -R_i(1) ~ 0 relative to its terms
    ones = np.ones(8)
    prod = m.rc0 + m.rc1 @ ones + np.array([sum(m.rc2[i, t] for t in range(len(m.bil))) for i in range(8)])
    scale = np.abs(m.rc1).sum(1) + np.abs(m.rc2).sum(1) + np.abs(m.rc0)
    assert np.all(np.abs(prod[:5]) <= 2e-3 * scale[:5])


def test_pore_as_published_drops_fluxes():
    a = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9, as_published=True).model
    b = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9).model
    assert not a.wall_flux.any() and not a.exit_kappa.any()
    assert np.count_nonzero(b.wall_flux) == 4 and np.count_nonzero(b.exit_kappa) == 8


def test_pore_quirks():
    assert pore_parameters(concentration_elec=0.5, L=50e-9, R=2.5e-9).mesh_name == "L_50_R_2.xml"  # Q4
    assert pore_parameters(concentration_elec=0.5, L=50e-9, R=7.5e-9).mesh_name == "L_50_R_7.xml"  # Q4
    with pytest.raises(KeyError):
        pore_parameters(concentration_elec=0.1)  # Q9: no C0_CO/C0_H2 in the 0.1 M file
    with pytest.raises(KeyError):
        pore_parameters(concentration_elec=0.5, cation="Cs")  # Q9: 0.5 M file is K-only
    with pytest.raises(FileNotFoundError):
        pore_parameters(concentration_elec=0.3)


def test_edl_scalars_survey_values():
    ep = edl_parameters()
    s = ep.scalars
    assert rel(s["L_debye"], 9.71397e-10) < 1e-6
    assert rel(s["q"], 1.0608928e9) < 1e-7
    assert rel(s["dt"], 0.190036) < 1e-5
    assert rel(s["J_OH"], -13786.32) < 1e-6
    assert rel(s["J_CO2"], 0.0318624) < 1e-5
    assert s["J_H"] == 0.0
    assert ep.mesh_name == "1D_variable_50um_mesh_5990.xml.gz" and ep.tot_num_steps == 100
    assert ep.species == ["H", "OH", "HCO3", "CO32", "CO2", "K"]
    assert rel(ep.model.inv_dt, 1.0 / (s["dt"] * ep.L_D)) < 1e-14
    cs = edl_parameters(L_n=1e-6, cation="Cs", voltage_multiplier=-10.0)
    assert cs.mesh_name == "1D_variable_1um_mesh_1090.xml.gz" and cs.n_water["Cs"] == 3
    assert cs.voltage_scaled == -10.0
    # steric caps: 1/(a^3 N_A)
    assert rel(1 / (0.662e-9 ** 3 * 6.022e23), 5723.8) < 1e-4 and rel(1 / (0.658e-9 ** 3 * 6.022e23), 5828.8) < 1e-4


def test_edl_quirks_and_staging():
    with pytest.raises(UnboundLocalError):
        edl_parameters(L_n=200e-6)  # Q8
    assert edl_parameters(mesh_structure="uniform").mesh_name == "1D_uniform_mesh_1000.xml.gz"  # file absent (Q8)
    full = edl_parameters(dry_run=False)
    assert full.stage_steps == [10000, 10000] and len(full.dts) == 2
    assert rel(full.dts[1] / full.dts[0], 100.0) < 1e-12
    h = edl_parameters(H_OHP=0.5)
    assert h.current_H_frac == 0.001
    JH, JOH = h.ohp_fluxes(0.001)
    assert rel(JH, h.J_H_prefactor * 10.0 * 0.001) < 1e-15 and JOH < 0
    pnp = edl_parameters(model="PNP")
    assert not pnp.model.steric


def test_sechenov():
    pp = pore_parameters(concentration_elec=0.5, L=50e-9, R=5e-9)
    v = pp.sechenov_co2_scaled(1.0, 1.0, 1.0, 1.0)
    b = pp.bulk_conc
    s = sum((pp.h_sechenov[k] + pp.h_sechenov["CO2_0"]) * b[k] / 1000 for k in ("OH", "HCO3", "CO32", "K"))
    lnK = 93.4517 * (100 / 298.15) - 60.2409 + 23.3585 * math.log(298.15 / 100)
    assert rel(v, 0.95 * math.exp(lnK) * 1000 * 10 ** (-s) / b["CO2"]) < 1e-14
    assert v < pp.eq_conc_CO2_scaled  # salting out lowers the solubility
    assert co2_conc(298.15, 0.95, {}, {"CO2_0": 0.0, "CO2_T": 0.0}) == pytest.approx(0.95 * math.exp(lnK) * 1000)


def test_rxn_diff_parameters_match_reference_formulas():
    """Scalars of reference 1D/rxn_diff_planar.py:151-246 (0.1 M, 50 um: the same OHP fluxes as the EDL script)."""
    from gmpnp_amd.rxndiff1d import rxn_diff_parameters
    rp = rxn_diff_parameters()
    assert rp.mesh_name == "1D_variable_50um_mesh_5990.xml.gz" and rp.num_steps == 500
    assert abs(rp.scalars["J_OH"] / -13786.32 - 1) < 1e-6 and abs(rp.scalars["J_CO2"] / 0.0318624 - 1) < 1e-5
    assert abs(rp.time_constant - (50e-6) ** 2 / rp.diff_coeff["CO32"]) < 1e-15
    assert abs(rp.dt * rp.time_constant - 2.0e-2) < 1e-15 and abs(rp.model.inv_dt * rp.dt - 1) < 1e-14
    m = rp.model
    assert not m.steric and not m.z.any() and not m.a.any() and m.point_flux[1] == rp.scalars["J_OH"]
    assert rp.species == ["H", "OH", "HCO3", "CO32", "CO2", "K"] and m.rc1[5].tolist() == [0.0] * 6  # cation: no source
    with pytest.raises(UnboundLocalError):
        rxn_diff_parameters(L_n=20e-6)  # no mesh for 20 um: the reference leaves mesh_number unbound


def test_column_medians_equals_numpy_median():
    """The per-step medians of the 3D driver (reference 3D:817-824) use one selection pass; values are np.median's."""
    from gmpnp_amd.solver import column_medians
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 10, 3679, 3680):
        v = rng.standard_normal((n, 9))
        assert np.array_equal(column_medians(v, (1, 2, 3, 7)), np.array([np.median(v[:, c]) for c in (1, 2, 3, 7)]))
    v[5, 2] = np.nan
    m = column_medians(v, (1, 2, 3, 7))
    assert np.isnan(m[1]) and not np.isnan(m[0])


def test_pore_dirichlet_cache_equals_a_fresh_merge(pore10):
    """bc4 is rebuilt every time step (reference 3D:835-838); the cached dof set must give what DOLFIN's in-order
    application gives for any CO2 value, and must notice a changed wall potential (sweep continuation)."""
    import copy
    from gmpnp_amd.problem import merge_dirichlet, pore_dirichlet
    pp, _, _, bnd = pore10
    pp = copy.copy(pp)

    def fresh(co2):
        s1, s2, s3 = bnd.dirichlet_vertices[1], bnd.dirichlet_vertices[2], bnd.dirichlet_vertices[3]
        ns = len(pp.species)
        return merge_dirichlet([(s1, ns, 0.0), (s3, ns, 0.0), (s2, ns, pp.voltage_scaled), (s1, 4, co2),
                                (s1, 5, pp.eq_conc_CO_scaled), (s1, 6, pp.eq_conc_H2_scaled)], ns + 1)

    for co2 in (5.7, 4.2, 5.7):
        d, v = pore_dirichlet(pp, bnd, co2)
        d0, v0 = fresh(co2)
        assert np.array_equal(d, d0) and np.array_equal(v, v0)
    pp.voltage_scaled = -3.0
    d, v = pore_dirichlet(pp, bnd, 1.5)
    d0, v0 = fresh(1.5)
    assert np.array_equal(d, d0) and np.array_equal(v, v0)
    v[:] = 0.0  # the caller's array is its own
    assert np.array_equal(pore_dirichlet(pp, bnd, 1.5)[1], v0)
