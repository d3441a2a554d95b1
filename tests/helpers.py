"""Shared helpers for the parity tests: golden loading and oracle plumbing.

The oracle (``oracle/``) is imported here because this is test infrastructure.
"""

from __future__ import annotations

import functools
import warnings
from pathlib import Path

import numpy as np
import torch

from mythos_amd.input import defaults, topology, trajectory
from oracle import oxdna_oracle as orc

GOLDEN = Path(__file__).resolve().parent / "golden"

SPLIT_COLUMNS = [
    "t",
    "fene",
    "bonded_excluded_volume",
    "stacking",
    "unbonded_excluded_volume",
    "hydrogen_bonding",
    "cross_stacking",
    "coaxial_stacking",
    "debye",
]
# tolerances the reference's own golden tests assert (dna2/tests/test_integration.py:94..374)
TERM_ATOL = {
    "fene": 1e-6,
    "bonded_excluded_volume": 1e-6,
    "stacking": 1e-6,
    "unbonded_excluded_volume": 1e-6,
    "hydrogen_bonding": 1e-3,
    "cross_stacking": 1e-3,
    "coaxial_stacking": 1e-6,
    "debye": 1e-3,
}


def model_dir(model: int) -> str:
    """1 -> dna1, 2 -> dna2, 3 -> rna2 (the model numbers of the C ABI)."""
    return {1: "dna1", 2: "dna2", 3: "rna2"}[model]


@functools.lru_cache(maxsize=None)
def load_golden(model: int, name: str, top_file: str = "generated.top"):
    base = GOLDEN / model_dir(model) / name
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        top = topology.from_oxdna_file(base / top_file)
    traj = trajectory.from_file(base / "output.dat", top.strand_counts, is_5p_3p=False)
    split = np.loadtxt(base / "split_energy.dat", skiprows=1)
    energy = np.loadtxt(base / "energy.dat")[1:, 1]
    return top, traj, split, energy


def read_ss_weights(path) -> dict:
    """oxDNA sequence-dependence file -> weight matrices (format: KEY = VALUE[f])."""
    pm = {}
    for line in Path(path).read_text().splitlines():
        kv = line.strip().replace(" ", "")
        if kv:
            k, v = kv.split("=")
            pm[k] = float(v.replace("f", ""))
    alpha = "ACGT"
    stack = np.array([[pm[f"STCK_{a}_{b}"] for b in alpha] for a in alpha])
    hb = np.zeros((4, 4))
    at = pm.get("HYDR_A_T", pm.get("HYDR_T_A"))
    gc = pm.get("HYDR_G_C", pm.get("HYDR_C_G"))
    hb[0, 3] = hb[3, 0] = at
    hb[2, 1] = hb[1, 2] = gc
    return {"eps_stack_kt_coeff": pm["STCK_FACT_EPS"], "ss_stack_weights": stack, "ss_hb_weights": hb}


def oracle_params(model: int, *, half_charged_ends=False, overrides=None, kt=None, salt=0.5):
    sim, cfg = defaults.default_configs_for(model_dir(model))
    for sec, d in (overrides or {}).items():
        cfg[sec].update(d)
    return orc.init_all(model, cfg, kt=sim["kT"] if kt is None else kt, salt_conc=salt, half_charged_ends=half_charged_ends)


def topo_tensors(top):
    return (
        torch.as_tensor(top.seq, dtype=torch.long),
        torch.as_tensor(top.is_end, dtype=torch.long),
        torch.as_tensor(top.bonded_neighbors, dtype=torch.long),
        torch.as_tensor(top.unbonded_neighbors, dtype=torch.long),
    )


def oracle_terms_traj(model, P, top, traj, frames=None, use_axes=True):
    """(S, n_terms) energies per nucleotide of the golden trajectory."""
    seq, is_end, b, u = topo_tensors(top)
    out = []
    quats = None if use_axes else traj.quaternions
    for f in range(len(traj.frames)) if frames is None else frames:
        c = torch.as_tensor(traj.center[f])
        if use_axes:
            e = orc.energy_terms(
                model, P, c, None, seq, is_end, b, u, box=traj.box_size,
                axes=(torch.as_tensor(traj.a1[f]), torch.as_tensor(traj.a3[f])),
            )
        else:
            e = orc.energy_terms(model, P, c, torch.as_tensor(quats[f]), seq, is_end, b, u, box=traj.box_size)
        out.append(e.numpy() / top.n_nucleotides)
    return np.array(out)


def read_pair_dat(path, n_blocks=None):
    """pair.dat -> list of dict{(i,j): row[9]} ; block k>=1 corresponds to frame k-1."""
    blocks, cur = [], None
    for line in Path(path).read_text().splitlines():
        if "#id1" in line:
            if n_blocks is not None and len(blocks) >= n_blocks and cur is not None:
                blocks.append(cur)
                cur = None
                break
            if cur is not None:
                blocks.append(cur)
            cur = {}
        elif line.strip() and not line.startswith("#") and cur is not None:
            tok = line.split()
            cur[(int(tok[0]), int(tok[1]))] = np.array([float(x) for x in tok[2:]])
    if cur is not None:
        blocks.append(cur)
    return blocks
