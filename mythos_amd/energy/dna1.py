"""oxDNA1 energy model with the reference's module surface (mythos/energy/dna1/__init__.py:22-123)."""

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
    Fene,
    FeneConfiguration,
    HydrogenBonding,
    HydrogenBondingConfiguration,
    Stacking,
    StackingConfiguration,
    UnbondedExcludedVolume,
    UnbondedExcludedVolumeConfiguration,
)
from mythos_amd.energy.terms import CoaxialStacking1 as CoaxialStacking
from mythos_amd.energy.terms import CoaxialStackingConfiguration1 as CoaxialStackingConfiguration
from mythos_amd.input import defaults


def default_configs() -> tuple[dict, dict]:
    """(simulation config, energy config) - mythos/energy/dna1/__init__.py:22-24."""
    return defaults.default_configs_for("dna1")


def default_energy_configs(overrides: dict = MappingProxyType({}), opts: dict = MappingProxyType({})) -> list[BaseConfiguration]:
    """mythos/energy/dna1/__init__.py:27-55."""
    sim, cfg = default_configs()

    def get_param(x):
        return {**cfg[x], **overrides.get(x, {})}

    def get_opts(x, dflt=BaseConfiguration.OPT_ALL):
        return opts.get(x, dflt)

    stacking_opts = tuple(set(cfg["stacking"].keys()) - {"kT", "ss_stack_weights"})
    return [
        FeneConfiguration.from_dict(get_param("fene"), get_opts("fene")),
        BondedExcludedVolumeConfiguration.from_dict(get_param("bonded_excluded_volume"), get_opts("bonded_excluded_volume")),
        StackingConfiguration.from_dict({**get_param("stacking"), "kt": overrides.get("kT", sim["kT"])}, get_opts("stacking", stacking_opts)),
        UnbondedExcludedVolumeConfiguration.from_dict(get_param("unbonded_excluded_volume"), get_opts("unbonded_excluded_volume")),
        HydrogenBondingConfiguration.from_dict(get_param("hydrogen_bonding"), get_opts("hydrogen_bonding")),
        CrossStackingConfiguration.from_dict(get_param("cross_stacking"), get_opts("cross_stacking")),
        CoaxialStackingConfiguration.from_dict(get_param("coaxial_stacking"), get_opts("coaxial_stacking")),
    ]


def default_energy_fns() -> list[type[BaseEnergyFunction]]:
    """mythos/energy/dna1/__init__.py:58-68."""
    return [Fene, BondedExcludedVolume, Stacking, UnbondedExcludedVolume, HydrogenBonding, CrossStacking, CoaxialStacking]


def default_transform_fn() -> Geometry:
    """Site geometry (mythos/energy/dna1/__init__.py:71-81)."""
    return Geometry(model=1, params=default_configs()[1]["geometry"])


def create_default_energy_fn(topology, displacement_fn=DEFAULT_DISPLACEMENT) -> EnergyFunction:
    """mythos/energy/dna1/__init__.py:84-102."""
    return ComposedEnergyFunction.from_lists(
        energy_fns=default_energy_fns(),
        energy_configs=default_energy_configs(),
        transform_fn=default_transform_fn(),
        displacement_fn=displacement_fn,
        topology=topology,
    )


__all__ = [
    "BondedExcludedVolume", "BondedExcludedVolumeConfiguration", "CoaxialStacking", "CoaxialStackingConfiguration",
    "CrossStacking", "CrossStackingConfiguration", "Fene", "FeneConfiguration", "HydrogenBonding",
    "HydrogenBondingConfiguration", "Stacking", "StackingConfiguration", "UnbondedExcludedVolume",
    "UnbondedExcludedVolumeConfiguration", "create_default_energy_fn", "default_configs", "default_energy_configs",
    "default_energy_fns", "default_transform_fn",
]
