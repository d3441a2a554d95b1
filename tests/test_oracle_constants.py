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


def test_oxrna2_constants_against_oxdnas_external_model_file():
    """data/test-data/regr-rna2-*/external_model.txt: oxRNA2's constants as an oxDNA external-model file, independent and
    dependent ones, 13 - 16 printed digits - the oxRNA2 counterpart of model_template.h, read by no reference test.  Every
    smoothing constant the oracle derives for oxRNA2 (and every independent constant it starts from) against it."""
    ext = {}
    for line in (H.GOLDEN / "regr" / "rna2-external-model" / "external_model.txt").read_text().splitlines():
        if "=" in line:
            k, v = line.split("=")
            ext[k.strip()] = float(v)
    P = H.oracle_params(3, half_charged_ends=False, salt=1.0)
    pairs = []  # (oracle value, name in the file)
    ne, st, hb, cr, cx = (P[k] for k in ("unbonded_excluded_volume", "stacking", "hydrogen_bonding", "cross_stacking", "coaxial_stacking"))
    for n, key in ((1, "backbone"), (2, "base"), (3, "back_base"), (4, "base_back")):
        pairs += [(ne[f"b_{key}"], f"RNA_EXCL_B{n}"), (ne[f"dr_c_{key}"], f"RNA_EXCL_RC{n}"), (ne[f"sigma_{key}"], f"RNA_EXCL_S{n}"),
                  (ne[f"dr_star_{key}"], f"RNA_EXCL_R{n}")]
    for p, pre, suf in ((st, "RNA_STCK", "stack"), (hb, "RNA_HYDR", "hb"), (cr, "RNA_CRST", "cross"), (cx, "RNA_CXST", "coax")):
        pairs += [(p[f"b_low_{suf}"], f"{pre}_BLOW"), (p[f"dr_c_low_{suf}"], f"{pre}_RCLOW"), (p[f"b_high_{suf}"], f"{pre}_BHIGH"),
                  (p[f"dr_c_high_{suf}"], f"{pre}_RCHIGH"), (p[f"dr_low_{suf}"], f"{pre}_RLOW"), (p[f"dr_high_{suf}"], f"{pre}_RHIGH")]
    for k in (5, 6):
        pairs += [(st[f"b_stack_{k}"], f"RNA_STCK_THETA{k}_B"), (st[f"delta_theta_stack_{k}_c"], f"RNA_STCK_THETA{k}_TC")]
    for k, name in ((9, "THETAB1"), (10, "THETAB2")):  # the two backbone-direction angles of oxRNA's stacking
        pairs += [(st[f"b_stack_{k}"], f"STCK_{name}_B"), (st[f"delta_theta_stack_{k}_c"], f"STCK_{name}_TC"), (st[f"a_stack_{k}"], f"STCK_{name}_A")]
    for k in (1, 2):
        pairs += [(st[f"b_neg_cos_phi{k}_stack"], f"RNA_STCK_PHI{k}_B"), (st[f"neg_cos_phi{k}_c_stack"], f"RNA_STCK_PHI{k}_XC")]
    for k in (1, 2, 3, 4, 7, 8):
        pairs += [(hb[f"b_hb_{k}"], f"RNA_HYDR_THETA{k}_B"), (hb[f"delta_theta_hb_{k}_c"], f"RNA_HYDR_THETA{k}_TC")]
    for k in (1, 2, 3, 7, 8):  # (the file spells the cross-stacking B constants RNA_THETAk_B)
        pairs += [(cr[f"b_cross_{k}"], f"RNA_THETA{k}_B"), (cr[f"delta_theta_cross_{k}_c"], f"RNA_CRST_THETA{k}_TC"), (cr[f"theta0_cross_{k}"], f"RNA_CRST_THETA{k}_T0")]
    for k in (1, 4, 5, 6):
        pairs += [(cx[f"b_coax_{k}"], f"RNA_CXST_THETA{k}_B"), (cx[f"delta_theta_coax_{k}_c"], f"RNA_CXST_THETA{k}_TC")]
    for k in (3, 4):
        pairs += [(cx[f"b_cos_phi{k}_coax"], f"RNA_CXST_PHI{k}_B"), (cx[f"cos_phi{k}_c_coax"], f"RNA_CXST_PHI{k}_XC")]
    pairs += [(st["eps_stack_base"], "RNA_STCK_BASE_EPS"), (st["eps_stack_kt_coeff"], "RNA_STCK_FACT_EPS"), (hb["eps_hb"], "RNA_HYDR_EPS"),
              (cr["k_cross"], "RNA_CRST_K"), (cx["k_coax"], "RNA_CXST_K_OXDNA"), (P["fene"]["r0_backbone"], "RNA_FENE_R0"),
              (cx["theta0_coax_1"], "RNA_CXST_THETA1_T0_OXDNA"), (st["dr0_stack"], "RNA_STCK_R0"), (st["dr_c_stack"], "RNA_STCK_RC")]
    g = P["geometry"]
    pairs += [(g[k], n) for k, n in (("pos_stack_3_a1", "RNA_POS_STACK_3_a1"), ("pos_stack_3_a2", "RNA_POS_STACK_3_a2"), ("pos_stack_5_a1", "RNA_POS_STACK_5_a1"),
                                     ("pos_stack_5_a2", "RNA_POS_STACK_5_a2"), ("pos_back_a1", "RNA_POS_BACK_a1"), ("pos_back_a3", "RNA_POS_BACK_a3"),
                                     ("p5_x", "p5_x"), ("p5_y", "p5_y"), ("p5_z", "p5_z"), ("p3_x", "p3_x"), ("p3_y", "p3_y"), ("p3_z", "p3_z"))]
    assert len(pairs) >= 110
    for mine, name in pairs:
        assert float(mine) == pytest.approx(ext[name], rel=1e-11, abs=1e-13), (name, float(mine), ext[name])
