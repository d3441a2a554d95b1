"""Dependent smoothing constants of the oracle vs oxDNA's literal model.h values
(fixture: tests/golden/oxdna_model_constants.json, from data/templates/model_template.h:24-242)."""

import json

import pytest

from tests import helpers as H

C = json.loads((H.GOLDEN / "oxdna_model_constants.json").read_text())


def _close(a, b):
    # model.h prints 6 significant digits
    assert float(a) == pytest.approx(b, rel=2e-5, abs=1e-6), (float(a), b)


def test_excluded_volume_constants():
    p = H.oracle_params(1)["unbonded_excluded_volume"]
    for n, key in ((1, "backbone"), (2, "base"), (3, "back_base"), (4, "base_back")):
        _close(p[f"b_{key}"], C[f"EXCL_B{n}"])
        _close(p[f"dr_c_{key}"], C[f"EXCL_RC{n}"])


def test_f1_f2_radial_constants():
    P = H.oracle_params(1)
    hb, st, cr, cx = P["hydrogen_bonding"], P["stacking"], P["cross_stacking"], P["coaxial_stacking"]
    for p, pre, suf in ((hb, "HYDR", "hb"), (st, "STCK", "stack"), (cr, "CRST", "cross"), (cx, "CXST", "coax")):
        _close(p[f"b_low_{suf}"], C[f"{pre}_BLOW"])
        _close(p[f"b_high_{suf}"], C[f"{pre}_BHIGH"])
        _close(p[f"dr_c_low_{suf}"], C[f"{pre}_RCLOW"])
        _close(p[f"dr_c_high_{suf}"], C[f"{pre}_RCHIGH"])


@pytest.mark.parametrize(
    ("section", "prefix", "suffix", "ks"),
    [
        ("hydrogen_bonding", "HYDR", "hb", (1, 2, 3, 4, 7, 8)),
        ("stacking", "STCK", "stack", (4, 5, 6)),
        ("cross_stacking", "CRST", "cross", (1, 2, 3, 4, 7, 8)),
        ("coaxial_stacking", "CXST", "coax", (1, 4, 5, 6)),
    ],
)
def test_f4_constants(section, prefix, suffix, ks):
    p = H.oracle_params(1)[section]
    for k in ks:
        _close(p[f"b_{suffix}_{k}"], C[f"{prefix}_THETA{k}_B"])
        _close(p[f"delta_theta_{suffix}_{k}_c"], C[f"{prefix}_THETA{k}_TC"])


def test_f5_constants():
    P = H.oracle_params(1)
    _close(P["stacking"]["b_neg_cos_phi1_stack"], C["STCK_PHI1_B"])
    _close(P["stacking"]["neg_cos_phi1_c_stack"], C["STCK_PHI1_XC"])
    _close(P["coaxial_stacking"]["b_cos_phi3_coax"], C["CXST_PHI3_B"])
    _close(P["coaxial_stacking"]["cos_phi3_c_coax"], C["CXST_PHI3_XC"])


def test_debye_constants_at_default_conditions():
    """lambda 0.508152, r_high 1.524455, r_cut 2.286682, B 3.05192e-3."""
    d = H.oracle_params(2)["debye"]
    assert float(d["lambda_"]) == pytest.approx(0.508152, rel=1e-5)
    assert float(d["r_high"]) == pytest.approx(1.524455, rel=1e-5)
    assert float(d["r_cut"]) == pytest.approx(2.286682, rel=1e-5)
    assert float(d["smoothing_coeff"]) == pytest.approx(3.05192e-3, rel=1e-5)
    assert float(d["prefactor"]) == pytest.approx(0.0542925, rel=1e-5)
