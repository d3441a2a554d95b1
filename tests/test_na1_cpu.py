"""oxNA host side (no GPU): the na1 configurations, the three flat vectors, what goes together.

Reference: mythos/energy/na1/*.py, mythos/input/na1/default_energy.toml, na1/tests/test_integration.py:104-141."""

import numpy as np
import pytest
import torch

from mythos_amd import _lib
from mythos_amd.energy import dna2, na1, rna2
from mythos_amd.energy import flat_params as fp
from mythos_amd.energy import terms as T
from mythos_amd.input import defaults
from tests import helpers as H


def test_configurations_carry_three_prefixed_sets():
    top, _, _, is_rna = H.load_golden_na1("simple-helix-dna-rna")
    assert is_rna.tolist() == [False] * 8 + [True] * 8
    sim, merged = na1.default_configs()
    assert merged["fene"]["dna_r0_backbone"] == 0.7564 and merged["fene"]["rna_r0_backbone"] == 0.761070781051
    assert "drh_eps_backbone" not in merged["fene"] and merged["hydrogen_bonding"]["drh_eps_hb"] == 1.5
    assert merged["cross_stacking"]["drh_k_cross"] == 44.535 and "rna_theta0_cross_4" not in merged["cross_stacking"]
    cfgs = {type(c).__name__: c for c in na1.default_energy_configs(top.nt_type)}
    hb = cfgs["HydrogenBondingConfiguration"].init_params()
    for which, eps in (("dna", 1.0678), ("rna", 0.870439), ("drh", 1.5)):
        sub = hb[f"{which}_config"]
        assert isinstance(sub, T.HydrogenBondingConfiguration) and float(sub["eps_hb"]) == eps
        direct = T.HydrogenBondingConfiguration(**{k[4:]: v for k, v in merged["hydrogen_bonding"].items() if k.startswith(which)}).init_params()
        for k in T.HydrogenBondingConfiguration.dependent_params:
            np.testing.assert_allclose(np.asarray(sub[k], dtype=np.float64), np.asarray(direct[k], dtype=np.float64), rtol=0, atol=0)
    st = cfgs["StackingConfiguration"].init_params()
    assert isinstance(st["rna_config"], T.StackingConfigurationRna2) and isinstance(st["dna_config"], T.StackingConfiguration)
    assert float(st["dna_config"]["kt"]) == sim["kT"] == float(st["rna_config"]["kt"])
    cx = cfgs["CoaxialStackingConfiguration"].init_params()
    assert isinstance(cx["dna_config"], T.CoaxialStackingConfiguration2) and isinstance(cx["drh_config"], T.CoaxialStackingConfiguration1)
    with pytest.raises(ValueError, match="nt_type"):
        na1.FeneConfiguration(**{k: v for k, v in merged["fene"].items()})
    # optimisable: everything but the types and the shared conditions
    assert "nt_type" not in cfgs["FeneConfiguration"].opt_params and len(cfgs["FeneConfiguration"].opt_params) == 10
    assert "kt" not in cfgs["StackingConfiguration"].opt_params and "rna_a_stack_9" in cfgs["StackingConfiguration"].opt_params
    assert set(cfgs["DebyeConfiguration"].opt_params) == {f"{w}_{n}" for w in ("dna", "rna", "drh") for n in ("q_eff", "lambda_factor", "prefactor_coeff")}


def test_three_flat_vectors():
    sim, cfg = defaults.default_configs_for("na1")
    named = fp.derive_flat_na1(cfg["dna"], cfg["rna"], cfg["drh"], kt=sim["kT"], salt_conc=0.5, half_charged_ends=False)
    names = _lib.param_names()
    flat = fp.pack_flat_na1(named, names)
    assert flat.shape == (3 * len(names),) and torch.isfinite(flat).all()
    sim2, cfg2 = defaults.default_configs_for("dna2")
    assert torch.equal(flat[: len(names)], fp.pack_flat(fp.derive_flat(2, cfg2, kt=sim2["kT"], salt_conc=0.5, half_charged_ends=False), names))
    sim3, cfg3 = defaults.default_configs_for("rna2")
    assert torch.equal(flat[len(names): 2 * len(names)], fp.pack_flat(fp.derive_flat(3, cfg3, kt=sim3["kT"], salt_conc=0.5, half_charged_ends=False), names))
    drh = named["drh"]
    assert float(drh["CRST_K"]) == 44.535 and float(drh["CRST_TH4_A"]) == 1.5  # oxDNA form: the theta4 block is live
    assert float(drh["CXST_PHI3_A"]) == 2.0 and float(drh["CXST_F6_A"]) == 0.0 and float(drh["DH_PREFACTOR"]) > 0  # f5 form + Debye
    assert float(drh["HYDR_EPS_03"]) == 1.5


def test_terms_and_geometry_go_together():
    top, _, _, _ = H.load_golden_na1("simple-helix-dna-rna")
    ef = na1.create_default_energy_fn(top)
    assert len(ef.energy_fns) == 8 and all(fn.model == 4 for fn in ef.energy_fns)
    T.check_term_models(4, ef.energy_fns)
    with pytest.raises(ValueError, match="go together"):
        T.check_term_models(2, ef.energy_fns)
    with pytest.raises(ValueError, match="go together"):
        T.check_term_models(4, dna2.create_default_energy_fn(top).energy_fns)
    with pytest.raises(ValueError, match="go together"):
        T.check_term_models(4, [*ef.energy_fns[:7], rna2.create_default_energy_fn(top).energy_fns[2]])
    g = defaults.default_configs_for("na1")[1]
    geo = na1.HybridNucleotide.geometry(
        dna_com_to_backbone_x=g["dna"]["geometry"]["com_to_backbone_x"], dna_com_to_backbone_y=g["dna"]["geometry"]["com_to_backbone_y"],
        dna_com_to_backbone_dna1=g["dna"]["geometry"]["com_to_backbone_dna1"], dna_com_to_hb=g["dna"]["geometry"]["com_to_hb"],
        dna_com_to_stacking=g["dna"]["geometry"]["com_to_stacking"], rna_com_to_backbone_x=g["rna"]["geometry"]["pos_back_a1"],
        rna_com_to_backbone_y=g["rna"]["geometry"]["pos_back_a3"], rna_com_to_hb=g["rna"]["geometry"]["pos_base"],
        rna_com_to_stacking=g["rna"]["geometry"]["pos_stack"], **{f"rna_{k}": g["rna"]["geometry"][k] for k in (
            "p3_x", "p3_y", "p3_z", "p5_x", "p5_y", "p5_z", "pos_stack_3_a1", "pos_stack_3_a2", "pos_stack_5_a1", "pos_stack_5_a2")})
    assert geo.model == 4 and geo.params["dna"] == na1.default_transform_fn().params["dna"]
    assert all(geo.params["rna"][k] == v for k, v in na1.default_transform_fn().params["rna"].items() if k in geo.params["rna"])
