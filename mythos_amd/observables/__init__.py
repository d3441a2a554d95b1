"""Per-frame observables used by DiffTRe objectives (a small part of mythos/observables/: SURVEY.md 8f-3)."""

from mythos_amd.observables.propeller import PropellerTwist

__all__ = ["PropellerTwist"]
