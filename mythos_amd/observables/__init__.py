"""Per-frame observables used by DiffTRe objectives (a small part of mythos/observables/: SURVEY.md 8f-3)."""

from mythos_amd.observables.base import get_duplex_quartets, local_helical_axis, nucleotide_sites
from mythos_amd.observables.persistence_length import PersistenceLength, persistence_length_fit, vector_autocorrelate
from mythos_amd.observables.pitch import PitchAngle, compute_pitch
from mythos_amd.observables.propeller import PropellerTwist
from mythos_amd.observables.rise import Rise

__all__ = ["PersistenceLength", "PitchAngle", "PropellerTwist", "Rise", "compute_pitch", "get_duplex_quartets", "local_helical_axis", "nucleotide_sites", "persistence_length_fit",
           "vector_autocorrelate"]
