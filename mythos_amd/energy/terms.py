"""oxDNA1 / oxDNA2 term classes and their configurations, with the reference's names.

Reference: mythos/energy/dna1/{fene,bonded_excluded_volume,stacking,unbonded_excluded_volume,
hydrogen_bonding,cross_stacking,coaxial_stacking}.py and mythos/energy/dna2/{stacking,
coaxial_stacking,debye}.py.  A configuration lists the required (independent) parameters in the
reference's order and derives the dependent ones in ``init_params`` through
``flat_params`` (the same closed forms the kernels' flat vector is built from); a term class binds a
configuration to a topology and evaluates through the fused HIP kernel (base.py).
"""

from __future__ import annotations

import torch

from mythos_amd.energy import flat_params as fp
from mythos_amd.energy.base import BaseEnergyFunction
from mythos_amd.energy.configuration import BaseConfiguration
from mythos_amd.input import defaults

F64 = torch.float64


def default_kt() -> float:
    return defaults.DNA2_SIMULATION["kT"]


def _blocks(prefix_map: dict, named: dict) -> dict:
    return {ref: named[flat] for ref, flat in prefix_map.items()}


def _f4_names(term: str, flat_prefix: str, ks) -> dict:
    out = {}
    for k in ks:
        out[f"b_{term}_{k}"] = f"{flat_prefix}_TH{k}_B"
        out[f"delta_theta_{term}_{k}_c"] = f"{flat_prefix}_TH{k}_TC"
    return out


def _radial_names(term: str, flat_prefix: str) -> dict:
    return {
        f"b_low_{term}": f"{flat_prefix}_BLOW",
        f"dr_c_low_{term}": f"{flat_prefix}_RCLOW",
        f"b_high_{term}": f"{flat_prefix}_BHIGH",
        f"dr_c_high_{term}": f"{flat_prefix}_RCHIGH",
    }


def _derive_with(model: int, section: str, cfg: BaseConfiguration, extra: dict | None = None) -> dict:
    """Run the flat derivation with this section's values and defaults elsewhere."""
    sections = {section: {k: cfg[k] for k in (*type(cfg).required_params, *type(cfg).optional_params)
                          if k not in ("pseq", "pseq_constraints")}}  # the sequence distribution is not a flat parameter
    fill_missing_sections(model, sections)
    kw = dict(kt=default_kt())
    kw.update(extra or {})
    return fp.derive_flat(model, sections, **kw)


MODEL_NAMES = {1: "dna1", 2: "dna2", 3: "rna2"}  # the model numbers of the C ABI (include/mythos_hip.h)


def check_term_models(model: int, energy_fns) -> None:
    """The site geometry decides which instantiation of the kernels runs; a term class written for another functional
    form must not be evaluated by it silently."""
    if model == 4 or any(fn.model == 4 for fn in energy_fns):
        if model != 4 or any(fn.model != 4 for fn in energy_fns):
            raise ValueError("the oxNA terms (mythos_amd.energy.na1) and the oxNA geometry (HybridNucleotide) go together: "
                             "every term of the composed function must be an na1 term")
        return
    if any(fn.model > model for fn in energy_fns):
        raise ValueError("an oxDNA2-only term (Debye / dna2 stacking / dna2 coaxial) needs the oxDNA2 geometry, an oxRNA2 "
                         "term (rna2 stacking / cross-stacking) the oxRNA2 geometry")
    if model == 3:
        for fn in energy_fns:
            if fn.term in ("stacking", "cross_stacking") and fn.model != 3:
                raise ValueError(f"the oxRNA2 geometry needs the rna2 {fn.term} term (mythos_amd.energy.rna2), got an oxDNA one")
            if fn.term == "coaxial_stacking" and fn.model != 1:
                raise ValueError("oxRNA2 uses the oxDNA1 form of the coaxial term (dna1.CoaxialStacking)")


def fill_missing_sections_na1(sets: dict) -> None:
    """oxNA: the three sets of sections (DNA-DNA, RNA-RNA, hybrid); whatever a composed function does not carry comes from
    the defaults (weight 0, so it only has to be well-formed)."""
    _, cfg = defaults.default_configs_for("na1")
    for which, sections in cfg.items():
        for sec, vals in sections.items():
            sets[which].setdefault(sec, vals)


def fill_missing_sections(model: int, sections: dict) -> None:
    """Terms absent from a composed function still need well-formed (unused, weight 0) parameters."""
    missing = [sec for sec in defaults.energy_section_names(MODEL_NAMES[model]) if sec not in sections]
    if missing:  # (a copy of the defaults only when one is needed: a composed function of all terms has none missing)
        _, cfg = defaults.default_configs_for(MODEL_NAMES[model])
        for sec in missing:
            sections[sec] = cfg[sec]
    if model == 2 and "debye" in sections:
        sections["debye"] = dict(sections["debye"])
    if model == 1 and "com_to_backbone" not in sections["geometry"]:
        # dna1 terms evaluated with a dna2 geometry object never happens; guard for clarity
        raise ValueError("oxDNA1 terms need the oxDNA1 geometry (com_to_backbone)")


# ------------------------------------------------------------------------------------------------
# configurations
# ------------------------------------------------------------------------------------------------
class FeneConfiguration(BaseConfiguration):
    """dna1/fene.py:16-28."""

    required_params = ("eps_backbone", "r0_backbone", "delta_backbone", "fmax", "finf")
    _derive = staticmethod(lambda self: {})


def _exc_names(with_backbone: bool, prefix: str) -> dict:
    m = {}
    for key, name in (("base", "BASE"), ("back_base", "BACK_BASE"), ("base_back", "BASE_BACK")) + (
        (("backbone", "BACKBONE"),) if with_backbone else ()
    ):
        m[f"b_{key}"] = f"{prefix}_{name}_B"
        m[f"dr_c_{key}"] = f"{prefix}_{name}_RC"
    return m


class BondedExcludedVolumeConfiguration(BaseConfiguration):
    """dna1/bonded_excluded_volume.py:14-75."""

    required_params = ("eps_exc", "dr_star_base", "sigma_base", "sigma_back_base", "sigma_base_back",
                       "dr_star_back_base", "dr_star_base_back")
    dependent_params = ("b_base", "dr_c_base", "b_back_base", "dr_c_back_base", "b_base_back", "dr_c_base_back")
    _derive = staticmethod(lambda self: _blocks(_exc_names(False, "BEXC"), _derive_with(1, "bonded_excluded_volume", self)))


class UnbondedExcludedVolumeConfiguration(BaseConfiguration):
    """dna1/unbonded_excluded_volume.py:17-96."""

    required_params = ("eps_exc", "dr_star_base", "sigma_base", "dr_star_back_base", "sigma_back_base",
                       "dr_star_base_back", "sigma_base_back", "dr_star_backbone", "sigma_backbone")
    dependent_params = ("b_base", "dr_c_base", "b_back_base", "dr_c_back_base", "b_base_back", "dr_c_base_back",
                        "b_backbone", "dr_c_backbone")
    _derive = staticmethod(lambda self: _blocks(_exc_names(True, "NEXC"), _derive_with(1, "unbonded_excluded_volume", self)))


def _stacking_derive(self) -> dict:
    if self["pseq"] is not None and self["pseq_constraints"] is None:
        raise ValueError("pseq_constraints must be provided when pseq is provided.")  # dna1/stacking.py:121-122, hydrogen_bonding.py:149-150
    named = _derive_with(1, "stacking", self, {"kt": self["kt"]})
    m = {**_radial_names("stack", "STCK"), **_f4_names("stack", "STCK", (4, 5, 6))}
    for k in (1, 2):
        m[f"b_neg_cos_phi{k}_stack"] = f"STCK_PHI{k}_B"
        m[f"neg_cos_phi{k}_c_stack"] = f"STCK_PHI{k}_XC"
    out = _blocks(m, named)
    out["eps_stack"] = torch.stack([torch.stack([named[f"STCK_EPS_{i}{j}"] for j in range(4)]) for i in range(4)])
    return out


class StackingConfiguration(BaseConfiguration):
    """dna1/stacking.py:45-183 (shared by oxDNA2)."""

    required_params = (
        "eps_stack_base", "eps_stack_kt_coeff", "dr_low_stack", "dr_high_stack", "a_stack", "dr0_stack", "dr_c_stack",
        "theta0_stack_4", "delta_theta_star_stack_4", "a_stack_4", "theta0_stack_5", "delta_theta_star_stack_5",
        "a_stack_5", "theta0_stack_6", "delta_theta_star_stack_6", "a_stack_6", "neg_cos_phi1_star_stack", "a_stack_1",
        "neg_cos_phi2_star_stack", "a_stack_2", "kt",
    )
    optional_params = ("ss_stack_weights", "pseq", "pseq_constraints")
    dependent_params = (
        "b_low_stack", "dr_c_low_stack", "b_high_stack", "dr_c_high_stack", "b_stack_4", "delta_theta_stack_4_c",
        "b_stack_5", "delta_theta_stack_5_c", "b_stack_6", "delta_theta_stack_6_c", "b_neg_cos_phi1_stack",
        "neg_cos_phi1_c_stack", "b_neg_cos_phi2_stack", "neg_cos_phi2_c_stack", "eps_stack",
    )
    _derive = staticmethod(_stacking_derive)


def _hb_derive(self) -> dict:
    if self["pseq"] is not None and self["pseq_constraints"] is None:
        raise ValueError("pseq_constraints must be provided when pseq is provided.")  # dna1/stacking.py:121-122, hydrogen_bonding.py:149-150
    named = _derive_with(1, "hydrogen_bonding", self)
    out = _blocks({**_radial_names("hb", "HYDR"), **_f4_names("hb", "HYDR", (1, 2, 3, 4, 7, 8))}, named)
    out["eps_hb_weights"] = torch.stack([torch.stack([named[f"HYDR_EPS_{i}{j}"] for j in range(4)]) for i in range(4)])
    return out


class HydrogenBondingConfiguration(BaseConfiguration):
    """dna1/hydrogen_bonding.py:28-223."""

    required_params = ("eps_hb", "a_hb", "dr0_hb", "dr_c_hb", "dr_low_hb", "dr_high_hb") + tuple(
        n for k in (1, 2, 3, 4, 7, 8) for n in (f"a_hb_{k}", f"theta0_hb_{k}", f"delta_theta_star_hb_{k}")
    )
    optional_params = ("ss_hb_weights", "pseq", "pseq_constraints")
    dependent_params = ("b_low_hb", "dr_c_low_hb", "b_high_hb", "dr_c_high_hb") + tuple(
        n for k in (1, 2, 3, 4, 7, 8) for n in (f"b_hb_{k}", f"delta_theta_hb_{k}_c")
    ) + ("eps_hb_weights",)
    _derive = staticmethod(_hb_derive)


class CrossStackingConfiguration(BaseConfiguration):
    """dna1/cross_stacking.py:17-183."""

    required_params = ("dr_low_cross", "dr_high_cross", "k_cross", "r0_cross", "dr_c_cross") + tuple(
        n for k in (1, 2, 3, 4, 7, 8) for n in (f"theta0_cross_{k}", f"delta_theta_star_cross_{k}", f"a_cross_{k}")
    )
    dependent_params = ("b_low_cross", "dr_c_low_cross", "b_high_cross", "dr_c_high_cross") + tuple(
        n for k in (1, 2, 3, 4, 7, 8) for n in (f"b_cross_{k}", f"delta_theta_cross_{k}_c")
    )
    _derive = staticmethod(
        lambda self: _blocks({**_radial_names("cross", "CRST"), **_f4_names("cross", "CRST", (1, 2, 3, 4, 7, 8))},
                             _derive_with(1, "cross_stacking", self))
    )


_COAX_COMMON = ("dr_low_coax", "dr_high_coax", "k_coax", "dr0_coax", "dr_c_coax") + tuple(
    n for k in (4, 1, 5, 6) for n in (f"theta0_coax_{k}", f"delta_theta_star_coax_{k}", f"a_coax_{k}")
)
_COAX_DEP = ("b_low_coax", "dr_c_low_coax", "b_high_coax", "dr_c_high_coax") + tuple(
    n for k in (4, 1, 5, 6) for n in (f"b_coax_{k}", f"delta_theta_coax_{k}_c")
)


def _coax_map() -> dict:
    return {**_radial_names("coax", "CXST"), **_f4_names("coax", "CXST", (4, 1, 5, 6))}


class CoaxialStackingConfiguration1(BaseConfiguration):
    """dna1/coaxial_stacking.py:17-172."""

    required_params = _COAX_COMMON + ("cos_phi3_star_coax", "a_coax_3p", "cos_phi4_star_coax", "a_coax_4p")
    dependent_params = _COAX_DEP + ("b_cos_phi3_coax", "cos_phi3_c_coax", "b_cos_phi4_coax", "cos_phi4_c_coax")
    _derive = staticmethod(
        lambda self: _blocks(
            {**_coax_map(), "b_cos_phi3_coax": "CXST_PHI3_B", "cos_phi3_c_coax": "CXST_PHI3_XC",
             "b_cos_phi4_coax": "CXST_PHI4_B", "cos_phi4_c_coax": "CXST_PHI4_XC"},
            _derive_with(1, "coaxial_stacking", self),
        )
    )


class CoaxialStackingConfiguration2(BaseConfiguration):
    """dna2/coaxial_stacking.py:17-130: the dependents are set by init_params but not declared."""

    required_params = _COAX_COMMON + ("a_coax_1_f6", "b_coax_1_f6")
    hidden_dependent_params = _COAX_DEP
    _derive = staticmethod(lambda self: _blocks(_coax_map(), _derive_with(2, "coaxial_stacking", self)))


def _debye_derive(self) -> dict:
    named = _derive_with(2, "debye", self, {"kt": self["kt"], "salt_conc": self["salt_conc"],
                                            "half_charged_ends": bool(self["half_charged_ends"])})
    lam = 1.0 / named["DH_KAPPA"]
    return {"lambda_": lam, "kappa": named["DH_KAPPA"], "r_high": named["DH_RHIGH"], "prefactor": named["DH_PREFACTOR"],
            "smoothing_coeff": named["DH_BSMOOTH"], "r_cut": named["DH_RCUT"]}


class DebyeConfiguration(BaseConfiguration):
    """dna2/debye.py:15-64."""

    required_params = ("q_eff", "lambda_factor", "prefactor_coeff", "kt", "salt_conc", "half_charged_ends")
    hidden_dependent_params = ("lambda_", "kappa", "r_high", "prefactor", "smoothing_coeff", "r_cut")
    _derive = staticmethod(_debye_derive)


# ------------------------------------------------------------------------------------------------
# term classes (one per reference class; dna2 variants differ in site geometry / functional form)
# ------------------------------------------------------------------------------------------------
class Fene(BaseEnergyFunction):
    term = "fene"


class BondedExcludedVolume(BaseEnergyFunction):
    term = "bonded_excluded_volume"


class Stacking(BaseEnergyFunction):
    term = "stacking"


class UnbondedExcludedVolume(BaseEnergyFunction):
    term = "unbonded_excluded_volume"


class HydrogenBonding(BaseEnergyFunction):
    term = "hydrogen_bonding"


class CrossStacking(BaseEnergyFunction):
    term = "cross_stacking"


class CoaxialStacking1(BaseEnergyFunction):
    term = "coaxial_stacking"
    model = 1


class CoaxialStacking2(BaseEnergyFunction):
    term = "coaxial_stacking"
    model = 2


class Stacking2(Stacking):
    """dna2/stacking.py:14-44: cos(phi) terms on the oxDNA1 backbone site."""

    model = 2


# ------------------------------------------------------------------------------------------------
# oxRNA2 (mythos/energy/rna2/): its own stacking and cross-stacking; everything else is composed from the oxDNA1 / oxDNA2
# classes above with the oxRNA2 site geometry (rna2/tests/test_integration.py:52-374)
# ------------------------------------------------------------------------------------------------
def _stacking_rna2_derive(self) -> dict:
    if self["pseq"] is not None and self["pseq_constraints"] is None:
        raise ValueError("pseq_constraints must be provided when pseq is provided.")  # rna2/stacking.py:121-122
    named = _derive_with(3, "stacking", self, {"kt": self["kt"]})
    m = {**_radial_names("stack", "STCK"), **_f4_names("stack", "STCK", (5, 6, 9, 10))}
    for k in (1, 2):
        m[f"b_neg_cos_phi{k}_stack"] = f"STCK_PHI{k}_B"
        m[f"neg_cos_phi{k}_c_stack"] = f"STCK_PHI{k}_XC"
    out = _blocks(m, named)
    out["eps_stack"] = torch.stack([torch.stack([named[f"STCK_EPS_{i}{j}"] for j in range(4)]) for i in range(4)])
    return out


class StackingConfigurationRna2(BaseConfiguration):
    """rna2/stacking.py:18-176: theta 5, 6, 9, 10 (no theta 4)."""

    required_params = (
        "eps_stack_base", "eps_stack_kt_coeff", "dr_low_stack", "dr_high_stack", "a_stack", "dr0_stack", "dr_c_stack",
    ) + tuple(n for k in (5, 6, 9, 10) for n in (f"theta0_stack_{k}", f"delta_theta_star_stack_{k}", f"a_stack_{k}")) + (
        "neg_cos_phi1_star_stack", "a_stack_1", "neg_cos_phi2_star_stack", "a_stack_2", "kt",
    )
    optional_params = ("ss_stack_weights", "pseq", "pseq_constraints")
    dependent_params = ("b_low_stack", "dr_c_low_stack", "b_high_stack", "dr_c_high_stack") + tuple(
        n for k in (5, 6, 9, 10) for n in (f"b_stack_{k}", f"delta_theta_stack_{k}_c")
    ) + ("b_neg_cos_phi1_stack", "neg_cos_phi1_c_stack", "b_neg_cos_phi2_stack", "neg_cos_phi2_c_stack", "eps_stack")
    _derive = staticmethod(_stacking_rna2_derive)


class CrossStackingConfigurationRna2(BaseConfiguration):
    """rna2/cross_stacking.py:17-147: no theta4 block."""

    required_params = ("dr_low_cross", "dr_high_cross", "k_cross", "r0_cross", "dr_c_cross") + tuple(
        n for k in (1, 2, 3, 7, 8) for n in (f"theta0_cross_{k}", f"delta_theta_star_cross_{k}", f"a_cross_{k}")
    )
    dependent_params = ("b_low_cross", "dr_c_low_cross", "b_high_cross", "dr_c_high_cross") + tuple(
        n for k in (1, 2, 3, 7, 8) for n in (f"b_cross_{k}", f"delta_theta_cross_{k}_c")
    )
    _derive = staticmethod(
        lambda self: _blocks({**_radial_names("cross", "CRST"), **_f4_names("cross", "CRST", (1, 2, 3, 7, 8))},
                             _derive_with(3, "cross_stacking", self))
    )


class StackingRna2(Stacking):
    """rna2/stacking.py:179-292: 5' / 3' stacking sites, theta9 / theta10 against the p3 / p5 vectors."""

    model = 3


class CrossStackingRna2(CrossStacking):
    """rna2/cross_stacking.py:150-238."""

    model = 3


class Debye(BaseEnergyFunction):
    """dna2/debye.py:67-115."""

    term = "debye"
    model = 2

    def __init__(self, **kw):
        super().__init__(**kw)
        if self.is_end is None:
            raise ValueError("is_end must be provided either through topology or directly.")
