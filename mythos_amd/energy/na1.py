"""oxNA energy model for hybrid DNA / RNA systems (mythos/energy/na1/__init__.py:1-31).

Every na1 term of the reference carries the types of the nucleotides (``nt_type``) and two or three parameter sets with
prefixed names - ``dna_*`` (oxDNA2 numbers), ``rna_*`` (oxRNA2), and for the unbonded terms ``drh_*`` (DNA-RNA hybrid
pairs, mythos/input/na1/default_energy.toml) - evaluates each set's term on all pairs and selects by the types of the
pair (e.g. na1/hydrogen_bonding.py:314-362).  Here the configurations have the same names and the same dependent
sub-configurations (``dna_config``, ``rna_config``, ``drh_config``); the selection happens per pair inside the HIP
energy kernel's oxNA instantiation (model 4 of the C ABI), which takes the three flat vectors and ``is_rna``
(and, for hydrogen bonding, the probabilistic sequence the reference's configuration carries).
Energies, forces and dU/dtheta come from that kernel; dynamics (HipMDSimulator / mythos_langevin_*) from the oxNA
instantiation of the fused MD step kernel.
"""

from __future__ import annotations

from types import MappingProxyType

import numpy as np

from mythos_amd.energy import terms as T
from mythos_amd.energy.base import (
    DEFAULT_DISPLACEMENT,
    BaseEnergyFunction,
    ComposedEnergyFunction,
    EnergyFunction,
    Geometry,
)
from mythos_amd.energy.configuration import BaseConfiguration
from mythos_amd.input import defaults

_SKIP = ("pseq", "pseq_constraints")  # never prefixed: the hydrogen-bonding configuration carries ONE distribution for its three sets


def _na1_configuration(name: str, doc: str, bases: dict, shared: tuple = (), pseq: bool = False) -> type:
    """A configuration whose required parameters are ``nt_type``, the shared ones, and the required parameters of each
    base configuration under its prefix; ``init_params`` builds the initialised base configurations."""
    req = ["nt_type", *shared]
    opt = []
    for which, cls in bases.items():
        req += [f"{which}_{n}" for n in cls.required_params if n not in shared]
        opt += [f"{which}_{n}" for n in cls.optional_params if n not in _SKIP]

    if pseq:  # one sequence distribution for all three sets (na1/hydrogen_bonding.py:127-128, 243-304)
        opt += list(_SKIP)

    def derive(self) -> dict:
        out = {}
        for which, cls in bases.items():
            vals = {n: (self[n] if n in shared else self[f"{which}_{n}"]) for n in cls.required_params}
            vals.update({n: self[f"{which}_{n}"] for n in cls.optional_params if n not in _SKIP and self[f"{which}_{n}"] is not None})
            if pseq and self["pseq"] is not None:
                vals.update(pseq=self["pseq"], pseq_constraints=self["pseq_constraints"])
            out[f"{which}_config"] = cls(**vals).init_params()
        return out

    def sections(self) -> dict:
        """{set: the section of that set in the reference's un-prefixed names} - what the flat vectors are derived from."""
        out = {}
        for which, cls in bases.items():
            sec = {n: (self[n] if n in shared else self[f"{which}_{n}"]) for n in cls.required_params}
            sec.update({n: self[f"{which}_{n}"] for n in cls.optional_params if n not in _SKIP})
            out[which] = sec
        return out

    return type(name, (BaseConfiguration,), {
        "__doc__": doc, "required_params": tuple(req), "optional_params": tuple(opt),
        "dependent_params": tuple(f"{w}_config" for w in bases), "non_optimizable_required_params": ("nt_type",),
        "_derive": staticmethod(derive), "sections": sections, "__module__": __name__,
    })


FeneConfiguration = _na1_configuration(
    "FeneConfiguration", "na1/fene.py:18-87.", {"dna": T.FeneConfiguration, "rna": T.FeneConfiguration})
BondedExcludedVolumeConfiguration = _na1_configuration(
    "BondedExcludedVolumeConfiguration", "na1/bonded_excluded_volume.py:17-93.",
    {"dna": T.BondedExcludedVolumeConfiguration, "rna": T.BondedExcludedVolumeConfiguration})
StackingConfiguration = _na1_configuration(
    "StackingConfiguration", "na1/stacking.py:20-190: the oxDNA2 stacking for DNA bonds, the oxRNA2 one for RNA bonds.",
    {"dna": T.StackingConfiguration, "rna": T.StackingConfigurationRna2}, shared=("kt",))
UnbondedExcludedVolumeConfiguration = _na1_configuration(
    "UnbondedExcludedVolumeConfiguration", "na1/unbonded_excluded_volume.py:17-137.",
    {w: T.UnbondedExcludedVolumeConfiguration for w in ("dna", "rna", "drh")})
HydrogenBondingConfiguration = _na1_configuration(
    "HydrogenBondingConfiguration", "na1/hydrogen_bonding.py:19-311 (the one na1 term that carries ``pseq`` / ``pseq_constraints``).",
    {w: T.HydrogenBondingConfiguration for w in ("dna", "rna", "drh")}, pseq=True)
CrossStackingConfiguration = _na1_configuration(
    "CrossStackingConfiguration", "na1/cross_stacking.py:19-259: oxDNA form for DNA-DNA and hybrid pairs, oxRNA2 form for RNA-RNA.",
    {"dna": T.CrossStackingConfiguration, "rna": T.CrossStackingConfigurationRna2, "drh": T.CrossStackingConfiguration})
CoaxialStackingConfiguration = _na1_configuration(
    "CoaxialStackingConfiguration", "na1/coaxial_stacking.py:19-246: oxDNA2 form for DNA-DNA, oxDNA1 form for RNA-RNA and hybrid pairs.",
    {"dna": T.CoaxialStackingConfiguration2, "rna": T.CoaxialStackingConfiguration1, "drh": T.CoaxialStackingConfiguration1})
DebyeConfiguration = _na1_configuration(
    "DebyeConfiguration", "na1/debye.py:18-100.", {w: T.DebyeConfiguration for w in ("dna", "rna", "drh")},
    shared=("kt", "salt_conc", "half_charged_ends"))


class _Na1Term(BaseEnergyFunction):
    model = 4


class Fene(_Na1Term):
    """na1/fene.py:80-107."""

    term = "fene"


class BondedExcludedVolume(_Na1Term):
    """na1/bonded_excluded_volume.py:90-116."""

    term = "bonded_excluded_volume"


class Stacking(_Na1Term):
    """na1/stacking.py:187-217."""

    term = "stacking"


class UnbondedExcludedVolume(_Na1Term):
    """na1/unbonded_excluded_volume.py:134-174."""

    term = "unbonded_excluded_volume"


class HydrogenBonding(_Na1Term):
    """na1/hydrogen_bonding.py:308-362."""

    term = "hydrogen_bonding"


class CrossStacking(_Na1Term):
    """na1/cross_stacking.py:256-300."""

    term = "cross_stacking"


class CoaxialStacking(_Na1Term):
    """na1/coaxial_stacking.py:243-287."""

    term = "coaxial_stacking"


class Debye(_Na1Term):
    """na1/debye.py:97-141."""

    term = "debye"

    def __init__(self, **kw):
        super().__init__(**kw)
        if self.is_end is None:
            raise ValueError("is_end must be provided either through topology or directly.")


class HybridNucleotide:
    """na1/nucleotide.py:12-78: the sites of every nucleotide in both geometries.  The kernels derive the sites of a
    nucleotide from (centre, quaternion) with the geometry of its own type, so what remains is the description."""

    @staticmethod
    def geometry(dna_com_to_backbone_x, dna_com_to_backbone_y, dna_com_to_backbone_dna1, dna_com_to_hb, dna_com_to_stacking,
                 rna_com_to_backbone_x, rna_com_to_backbone_y, rna_com_to_stacking, rna_com_to_hb, rna_p3_x, rna_p3_y, rna_p3_z,
                 rna_p5_x, rna_p5_y, rna_p5_z, rna_pos_stack_3_a1, rna_pos_stack_3_a2, rna_pos_stack_5_a1, rna_pos_stack_5_a2) -> Geometry:
        """The keyword arguments of the reference's ``HybridNucleotide.from_rigid_body`` (na1/nucleotide.py:23-47) -> the
        ``transform_fn`` of the energy functions."""
        return Geometry(model=4, params={
            "dna": {"com_to_backbone_x": dna_com_to_backbone_x, "com_to_backbone_y": dna_com_to_backbone_y,
                    "com_to_backbone_dna1": dna_com_to_backbone_dna1, "com_to_hb": dna_com_to_hb, "com_to_stacking": dna_com_to_stacking},
            "rna": {"pos_back_a1": rna_com_to_backbone_x, "pos_back_a3": rna_com_to_backbone_y, "pos_stack": rna_com_to_stacking,
                    "pos_base": rna_com_to_hb, "p3_x": rna_p3_x, "p3_y": rna_p3_y, "p3_z": rna_p3_z, "p5_x": rna_p5_x, "p5_y": rna_p5_y,
                    "p5_z": rna_p5_z, "pos_stack_3_a1": rna_pos_stack_3_a1, "pos_stack_3_a2": rna_pos_stack_3_a2,
                    "pos_stack_5_a1": rna_pos_stack_5_a1, "pos_stack_5_a2": rna_pos_stack_5_a2},
        })


def default_configs() -> tuple[dict, dict]:
    """(simulation config, merged energy sections with the prefixed names) - the ``merged_params`` of
    na1/tests/test_integration.py:104-141: oxDNA2 defaults under ``dna_``, oxRNA2 under ``rna_``, the hybrid toml under ``drh_``."""
    sim, cfg = defaults.default_configs_for("na1")
    merged: dict = {}
    for which, sections in cfg.items():
        for sec, vals in sections.items():
            merged.setdefault(sec, {}).update({f"{which}_{k}": v for k, v in vals.items()})
    return sim, merged


_CONFIGS = (("fene", FeneConfiguration), ("bonded_excluded_volume", BondedExcludedVolumeConfiguration), ("stacking", StackingConfiguration),
            ("unbonded_excluded_volume", UnbondedExcludedVolumeConfiguration), ("hydrogen_bonding", HydrogenBondingConfiguration),
            ("cross_stacking", CrossStackingConfiguration), ("coaxial_stacking", CoaxialStackingConfiguration), ("debye", DebyeConfiguration))


def default_energy_configs(nt_type, overrides: dict = MappingProxyType({}), opts: dict = MappingProxyType({})) -> list[BaseConfiguration]:
    sim, merged = default_configs()
    kt = overrides.get("kT", sim["kT"])
    shared = {"stacking": {"kt": kt},
              "debye": {"kt": kt, "salt_conc": overrides.get("salt_conc", sim["salt_conc"]),
                        "half_charged_ends": overrides.get("half_charged_ends", bool(sim["half_charged_ends"]))}}
    out = []
    for sec, cls in _CONFIGS:
        vals = {**merged[sec], **overrides.get(sec, {}), **shared.get(sec, {}), "nt_type": np.asarray(nt_type)}
        vals = {k: v for k, v in vals.items() if k in cls.required_params or k in cls.optional_params}
        # everything is optimisable but the shared conditions (kT, salt, end charges), as in dna2/__init__.py:45-52
        dflt = tuple(k for k in merged[sec] if k in cls.required_params) if sec in shared else BaseConfiguration.OPT_ALL
        out.append(cls.from_dict(vals, opts.get(sec, dflt)))
    return out


def default_energy_fns() -> list[type[BaseEnergyFunction]]:
    return [Fene, BondedExcludedVolume, Stacking, UnbondedExcludedVolume, HydrogenBonding, CrossStacking, CoaxialStacking, Debye]


def default_transform_fn() -> Geometry:
    _, cfg = defaults.default_configs_for("na1")
    return Geometry(model=4, params={"dna": cfg["dna"]["geometry"], "rna": cfg["rna"]["geometry"]})


def create_default_energy_fn(topology, displacement_fn=DEFAULT_DISPLACEMENT) -> EnergyFunction:
    return ComposedEnergyFunction.from_lists(
        energy_fns=default_energy_fns(),
        energy_configs=default_energy_configs(topology.nt_type),
        transform_fn=default_transform_fn(),
        displacement_fn=displacement_fn,
        topology=topology,
    )


__all__ = [
    "BondedExcludedVolume", "BondedExcludedVolumeConfiguration", "CoaxialStacking", "CoaxialStackingConfiguration", "CrossStacking",
    "CrossStackingConfiguration", "Debye", "DebyeConfiguration", "Fene", "FeneConfiguration", "HybridNucleotide", "HydrogenBonding",
    "HydrogenBondingConfiguration", "Stacking", "StackingConfiguration", "UnbondedExcludedVolume", "UnbondedExcludedVolumeConfiguration",
    "create_default_energy_fn", "default_configs", "default_energy_configs", "default_energy_fns", "default_transform_fn",
]
