"""Per-frame observables used by DiffTRe objectives (a small part of mythos/observables/: SURVEY.md 8f-3), evaluated
by the HIP library - stand-alone, or in the same call as the energy launch (``energy_fn.with_observables``)."""

from mythos_amd.observables.base import ObservableSet, get_duplex_quartets
from mythos_amd.observables.persistence_length import PersistenceLength, persistence_length_fit
from mythos_amd.observables.pitch import PitchAngle, compute_pitch
from mythos_amd.observables.propeller import PropellerTwist
from mythos_amd.observables.rise import Rise

__all__ = ["ObservableSet", "PersistenceLength", "PitchAngle", "PropellerTwist", "Rise", "compute_pitch", "get_duplex_quartets",
           "persistence_length_fit"]
