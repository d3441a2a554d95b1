"""Loss wrappers around observables (mirror of mythos/losses)."""
