"""Stern-layer post-processor (gmpnp_amd/stern.py; reference 1D/Stern_CO2ER.py).  No reference output of this script is held anywhere
(parity unpinned by data): the tests hold the integration against the closed form of the same ODE, the published quirks against their
arithmetic consequences, and the files against the names / keys the reference writes."""
import json
import os

import numpy as np
import pytest

from gmpnp_amd import stern

VT = stern.thermal_voltage()


def test_thermal_voltage_and_grid_are_the_references():
    assert abs(VT - 0.025683) < 2e-6                          # k_B T / e_0 at 298.15 K (parameters.yaml nat_const)
    x = stern.stern_grid(1.0e-11, -stern.L_STERN)
    assert x[0] == 0.0 and x[-1] == -stern.L_STERN and len(x) == abs(int(-stern.L_STERN / 1.0e-11)) and len(x) in (39, 40)
    xl = stern.stern_grid(1.0e-2, -stern.L_STERN * 1.0e9)
    assert len(xl) == abs(int(-stern.L_STERN * 1.0e9 / 1.0e-2)) and abs(xl[-1] + 0.4) < 1e-15


@pytest.mark.parametrize("v", sorted(stern.RECORDED))
def test_bdm_as_published_is_the_closed_form_with_the_roles_swapped(v):
    e, eps = stern.RECORDED[v]["E"], stern.RECORDED[v]["eps"]
    res = stern.stern(v, e, eps, model="BDM", as_published=True)
    x = res["x_nm"] * 1.0e-9
    pot, f = stern.bdm_closed_form(x, v * VT, -e, stern.EPS_REL_SURFACE, eps, stern.L_STERN)   # S1: 6 at the OHP, eps at the surface
    assert np.allclose(res["potential"], pot, rtol=0, atol=1e-12) and np.allclose(res["field"], -f, rtol=2e-5)
    # consequences: D = eps E is continuous, so the field at the surface is field_OHP * 6 / eps_OHP (it should be * eps_OHP / 6) ...
    assert abs(res["field_surf"] / (e * stern.EPS_REL_SURFACE / eps) - 1.0) < 3e-6
    # ... and (S2) the potential moves by field [V/nm] * metres: the "voltage at the electrode" is the OHP voltage to nine digits
    assert abs(res["voltage_electrode"] - res["voltage_OHP"]) < 4e-10 * abs(e)


def test_bdm_as_intended_raises_the_field_by_the_permittivity_ratio():
    v, e, eps = -5.0, stern.RECORDED[-5.0]["E"], stern.RECORDED[-5.0]["eps"]
    res = stern.stern(v, e, eps, model="BDM", as_published=False)
    pot, f = stern.bdm_closed_form(res["x_nm"], v * VT, -e, eps, stern.EPS_REL_SURFACE, stern.L_STERN * 1.0e9)
    assert np.allclose(res["potential"], pot, atol=1e-7) and np.allclose(res["field"], -f, rtol=2e-5)
    assert abs(res["field_surf"] / (e * eps / stern.EPS_REL_SURFACE) - 1.0) < 1e-5
    # the electrode sits below the OHP potential (cathodic field), by more than the constant-field estimate of the OHP field alone
    lin = stern.stern(v, e, eps, model="Stern_linear")
    assert res["voltage_electrode"] < lin["voltage_electrode"] < res["voltage_OHP"] < 0.0
    assert abs(lin["voltage_electrode"] - (v * VT + e * 0.4)) < 1e-15 and lin["field_surf"] == e


def test_files_keys_and_the_seven_metadata_lines(tmp_path, monkeypatch):
    monkeypatch.setenv("GMPNP_OUT", str(tmp_path))
    paths = stern.main(["--no_plots"])                        # no arguments: the five recorded triples (S3)
    assert len(paths) == 5 and len({os.path.dirname(p) for p in paths}) == 1
    for v, p in zip(stern.RECORDED, paths):
        assert p.endswith(os.path.join("_experiment", "voltage_scaled_OHP" + str(v)))
        un, sc = np.load(os.path.join(p, "stern_unscaled_BDM%s.npz" % v)), np.load(os.path.join(p, "stern_scaled_BDM%s.npz" % v))
        assert un.files == ["arr_0"] and sc.files == ["arr_0", "arr_1", "arr_2"] and un["arr_0"].shape == (len(sc["arr_0"]), 2)
        assert np.array_equal(sc["arr_1"], un["arr_0"][:, 0]) and np.array_equal(sc["arr_2"], -un["arr_0"][:, 1])
        lines = open(os.path.join(p, "metadata.txt")).read().splitlines()
        assert len(lines) == 7 and lines[0] == "model=BDM" and lines[2] == "field_OHP=%sV/nm" % stern.RECORDED[v]["E"]
        assert lines[3] == "Relative permittivity at the OHP is %s " % stern.RECORDED[v]["eps"] and lines[6] == "Stern length is 4e-10 m"
    one = stern.main(["--model", "Stern_linear", "--voltage_scaled_OHP", "-7.5", "--field_OHP", "-0.46", "--eps_rel_OHP", "50.2", "--no_plots"])
    assert len(one) == 1 and np.load(os.path.join(one[0], "stern_scaled_linear-7.5.npz")).files == ["arr_0", "arr_1"]


def test_the_ohp_values_come_out_of_a_run_directory(tmp_path, monkeypatch):
    monkeypatch.setenv("GMPNP_OUT", str(tmp_path))
    run = tmp_path / "run"
    run.mkdir()
    (run / "metadata.json").write_text(json.dumps({"voltage_multiplier": -10.0, "field_OHP": -0.6149631587776277, "eps_rel_OHP": 49.311548142969336}))
    (p,) = stern.main(["--from_run", str(run), "--no_plots"])
    assert "voltage_scaled_OHP-10.0" in p
    sc = np.load(os.path.join(p, "stern_scaled_BDM-10.0.npz"))
    assert np.array_equal(sc["arr_2"], stern.stern(-10.0, -0.6149631587776277, 49.311548142969336)["field"])
    with pytest.raises(ValueError):
        stern.stern(-1.0, -0.1, 70.0, model="nope")
