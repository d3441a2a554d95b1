"""dev: read one of the oxDNA runs under tests/golden/ (topology + trajectory) with the product's own readers - no oracle,
no test helpers (those are the checker's; a benchmark script measures the product)."""
import warnings
from pathlib import Path

import numpy as np

from mythos_amd.input import topology, trajectory

GOLDEN = Path(__file__).resolve().parent.parent / "tests" / "golden"


def load(model_dir: str, name: str, new_format: bool = False):
    """-> (topology, trajectory, is_rna (N,) bool)"""
    base = GOLDEN / model_dir / name
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        top = topology.from_oxdna_file(base / "generated.top")
    traj = trajectory.from_file(base / "output.dat", top.strand_counts, is_5p_3p=new_format)
    return top, traj, np.asarray(top.nt_type) == int(topology.NucleotideType.RNA)
