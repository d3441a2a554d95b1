"""DiffTRe reweighting math, objective protocol and the in-process optimisation loop (Adam)."""
