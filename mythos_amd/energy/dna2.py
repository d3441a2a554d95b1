"""oxDNA2 energy model with the reference's module surface (mythos/energy/dna2/__init__.py:28-146)."""

from __future__ import annotations

from types import MappingProxyType

from mythos_amd.energy.base import (
    DEFAULT_DISPLACEMENT,
    BaseEnergyFunction,
    ComposedEnergyFunction,
    EnergyFunction,
    Geometry,
)
from mythos_amd.energy.configuration import BaseConfiguration
from mythos_amd.energy.terms import (
    BondedExcludedVolume,
    BondedExcludedVolumeConfiguration,
    CrossStacking,
    CrossStackingConfiguration,
    Debye,
    DebyeConfiguration,
    Fene,
    FeneConfiguration,
    HydrogenBonding,
    HydrogenBondingConfiguration,
    StackingConfiguration,
    UnbondedExcludedVolume,
    UnbondedExcludedVolumeConfiguration,
)
from mythos_amd.energy.terms import CoaxialStacking2 as CoaxialStacking
from mythos_amd.energy.terms import CoaxialStackingConfiguration2 as CoaxialStackingConfiguration
from mythos_amd.energy.terms import Stacking2 as Stacking
from mythos_amd.input import defaults


def default_configs() -> tuple[dict, dict]:
    """mythos/energy/dna2/__init__.py:28-30."""
    return defaults.default_configs_for("dna2")


def default_energy_configs(overrides: dict = MappingProxyType({}), opts: dict = MappingProxyType({})) -> list[BaseConfiguration]:
    """mythos/energy/dna2/__init__.py:33-71."""
    sim, cfg = default_configs()

    def get_param(x):
        return {**cfg[x], **overrides.get(x, {})}

    def get_opts(x, dflt=BaseConfiguration.OPT_ALL):
        return opts.get(x, dflt)

    stacking_opts = tuple(set(cfg["stacking"].keys()) - {"kT", "ss_stack_weights"})
    debye_opts = tuple(set(cfg["debye"].keys()) - {"kT", "salt_conc"})
    debye_over = {
        "kt": overrides.get("kT", sim["kT"]),
        "salt_conc": overrides.get("salt_conc", sim["salt_conc"]),
        "half_charged_ends": overrides.get("half_charged_ends", bool(sim["half_charged_ends"])),
    }
    return [
        FeneConfiguration.from_dict(get_param("fene"), get_opts("fene")),
        BondedExcludedVolumeConfiguration.from_dict(get_param("bonded_excluded_volume"), get_opts("bonded_excluded_volume")),
        StackingConfiguration.from_dict({**get_param("stacking"), "kt": overrides.get("kT", sim["kT"])}, get_opts("stacking", stacking_opts)),
        UnbondedExcludedVolumeConfiguration.from_dict(get_param("unbonded_excluded_volume"), get_opts("unbonded_excluded_volume")),
        HydrogenBondingConfiguration.from_dict(get_param("hydrogen_bonding"), get_opts("hydrogen_bonding")),
        CrossStackingConfiguration.from_dict(get_param("cross_stacking"), get_opts("cross_stacking")),
        CoaxialStackingConfiguration.from_dict(get_param("coaxial_stacking"), get_opts("coaxial_stacking")),
        DebyeConfiguration.from_dict({**get_param("debye"), **debye_over}, get_opts("debye", debye_opts)),
    ]


def default_energy_fns() -> list[type[BaseEnergyFunction]]:
    """mythos/energy/dna2/__init__.py:74-85."""
    return [Fene, BondedExcludedVolume, Stacking, UnbondedExcludedVolume, HydrogenBonding, CrossStacking, CoaxialStacking, Debye]


def default_transform_fn() -> Geometry:
    """mythos/energy/dna2/__init__.py:88-99."""
    return Geometry(model=2, params=default_configs()[1]["geometry"])


def create_default_energy_fn(topology, displacement_fn=DEFAULT_DISPLACEMENT) -> EnergyFunction:
    """mythos/energy/dna2/__init__.py:102-120."""
    return ComposedEnergyFunction.from_lists(
        energy_fns=default_energy_fns(),
        energy_configs=default_energy_configs(),
        transform_fn=default_transform_fn(),
        displacement_fn=displacement_fn,
        topology=topology,
    )


__all__ = [
    "BondedExcludedVolume", "BondedExcludedVolumeConfiguration", "CoaxialStacking", "CoaxialStackingConfiguration",
    "CrossStacking", "CrossStackingConfiguration", "Debye", "DebyeConfiguration", "Fene", "FeneConfiguration",
    "HydrogenBonding", "HydrogenBondingConfiguration", "Stacking", "StackingConfiguration", "UnbondedExcludedVolume",
    "UnbondedExcludedVolumeConfiguration", "create_default_energy_fn", "default_configs", "default_energy_configs",
    "default_energy_fns", "default_transform_fn",
]
