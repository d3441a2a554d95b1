"""Units of the oxDNA model: the reference's conversion helpers (mythos/utils/units.py:5-38), same names and values.

kT in simulation units from a temperature - what every energy function, simulator and DiffTRe weight takes - plus the
length, force and energy scales.  The reference has no time unit; ``PS_PER_OXDNA_TIME`` is the conversion bench.py states
with its ns/day figure (one oxDNA time unit = 3.03 ps)."""

from __future__ import annotations

ANGSTROMS_PER_OXDNA_LENGTH = 8.518
ANGSTROMS_PER_NM = 10
NM_PER_OXDNA_LENGTH = ANGSTROMS_PER_OXDNA_LENGTH / ANGSTROMS_PER_NM
PN_PER_OXDNA_FORCE = 48.63
JOULES_PER_OXDNA_ENERGY = 4.142e-20
PS_PER_OXDNA_TIME = 3.03  # (not in the reference: bench.py's ns/day convention)


def get_kt(t_kelvin):
    """Temperature in Kelvin -> kT in simulation units (scalar or array)."""
    return 0.1 * t_kelvin / 300.0


def get_kt_from_c(t_celsius):
    """Temperature in Celsius -> kT in simulation units."""
    return get_kt(t_celsius + 273.15)


def get_kt_from_string(temp_str: str) -> float:
    """'300K' / '27C' (the spelling of an oxDNA input file's ``T``) -> kT in simulation units."""
    if temp_str.endswith("K"):
        return get_kt(float(temp_str.replace("K", "")))
    if temp_str.endswith("C"):
        return get_kt_from_c(float(temp_str.replace("C", "")))
    raise ValueError(f"Invalid temperature string: {temp_str}")


def from_kt(kt):
    """kT in simulation units -> temperature in Kelvin."""
    return 300.0 * kt / 0.1
