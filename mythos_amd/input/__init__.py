"""Host-side inputs: oxDNA topology / trajectory text formats, GROMACS readers, default parameter sets."""
