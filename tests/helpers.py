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
    """1 -> dna1, 2 -> dna2, 3 -> rna2, 4 -> na1 (the model numbers of the C ABI)."""
    return {1: "dna1", 2: "dna2", 3: "rna2", 4: "na1"}[model]


NA1_CASES = ("simple-helix-dna-dna", "simple-helix-rna-rna", "simple-helix-dna-rna", "simple-helix-rna-dna",
             "simple-coax-dna-dna-dna", "simple-coax-rna-rna-rna", "simple-coax-dna-dna-rna")
# the tolerances of mythos/energy/na1/tests/test_integration.py (:188,226,267,304,341,378,436,496)
NA1_TERM_ATOL = {"fene": 1e-6, "bonded_excluded_volume": 1e-6, "stacking": 1e-3, "unbonded_excluded_volume": 1e-6,
                 "hydrogen_bonding": 1e-4, "cross_stacking": 1e-4, "coaxial_stacking": 1e-6, "debye": 1e-5}


@functools.lru_cache(maxsize=None)
def load_golden_na1(name: str):
    """oxNA golden: new-format topology (5'->3', ``type=DNA|RNA`` per strand), trajectory read with is_5p_3p=True
    (na1/tests/test_integration.py:35-45).  -> (topology, trajectory, split energies, is_rna (N,) bool)."""
    base = GOLDEN / "na1" / name
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        top = topology.from_oxdna_file(base / "generated.top")
    traj = trajectory.from_file(base / "output.dat", top.strand_counts, is_5p_3p=True)
    split = np.loadtxt(base / "split_energy.dat", skiprows=1)
    return top, traj, split, np.asarray(top.nt_type) == int(topology.NucleotideType.RNA)


def oracle_params_na1(*, kt=None, salt=0.5, half_charged_ends=False):
    sim, cfg = defaults.default_configs_for("na1")
    return orc.init_all_na1(cfg["dna"], cfg["rna"], cfg["drh"], kt=sim["kT"] if kt is None else kt, salt_conc=salt,
                            half_charged_ends=half_charged_ends)


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


@functools.lru_cache(maxsize=None)
def random_dimers(model: int, bonded: bool, keep: int = 512, sample: int = 200_000, seed: int = 5):
    """A system of ``keep`` independent two-nucleotide molecules in random relative poses, selected from ``sample``
    random draws (with the oracle) so that every angular term of the pair is LIVE in many of them - the golden
    trajectories visit a narrow part of each term's angular window (both coaxial goldens hold 0 in every stored frame).

    bonded=False: two free nucleotides (unbonded excluded volume, H-bond, cross-stacking, coaxial stacking, Debye);
    bonded=True: a two-nucleotide strand whose backbone sites are within the FENE well (FENE, bonded excluded volume,
    stacking).  Returns (topology, center (2 keep, 3), quaternion (2 keep, 4), live-count per term name).
    model 4 (oxNA): every nucleotide is DNA or RNA at random (``topology.nt_type``), so the unbonded dimers cover DNA-DNA,
    RNA-RNA, DNA-RNA and RNA-DNA pairs - the hybrid coaxial term has no usable golden (na1/tests/test_integration.py:404).
    """
    rng = np.random.default_rng(seed + 10 * model + (1 if bonded else 0))
    m = sample
    q = rng.standard_normal((2 * m, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    c = np.zeros((2 * m, 3))
    d = rng.standard_normal((m, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    is_rna = np.zeros(2 * m, dtype=bool)
    if model == 4:
        P = oracle_params_na1()
        is_rna = rng.integers(0, 2, 2 * m).astype(bool)
        if bonded:  # a strand has one type (a chimeric bond is evaluated as DNA: one in eight dimers keeps mixed types)
            same = rng.random(m) < 0.875
            is_rna[1::2] = np.where(same, is_rna[0::2], is_rna[1::2])
    else:
        P = oracle_params(model, salt=1.0 if model == 3 else 0.5)
    if bonded:
        a1, a2, a3 = orc.quat_to_axes(torch.as_tensor(q))
        zero = torch.zeros(2 * m, 3, dtype=torch.float64)
        if model == 4:
            both = (is_rna[0::2] & is_rna[1::2]).repeat(2)[:, None]
            back = np.where(both, orc.Sites(3, P["rna"]["geometry"], zero, a1, a2, a3).back.numpy(),
                            orc.Sites(2, P["dna"]["geometry"], zero, a1, a2, a3).back.numpy())
            fene = [P["rna" if r else "dna"]["fene"] for r in (False, True)]
            r0 = np.where(both[0::2], float(fene[1]["r0_backbone"]), float(fene[0]["r0_backbone"]))
            delta = float(fene[0]["delta_backbone"])
        else:
            back = orc.Sites(model, P["geometry"], zero, a1, a2, a3).back.numpy()
            r0, delta = float(P["fene"]["r0_backbone"]), float(P["fene"]["delta_backbone"])
        c[1::2] = back[0::2] - back[1::2] + d * (r0 + rng.uniform(-0.6, 0.6, (m, 1)) * delta)
    else:
        c[1::2] = d * rng.uniform(0.3, 1.2, (m, 1))
    seq = rng.integers(0, 4, 2 * m)
    seq[1:m:2] = 3 - seq[0:m:2]  # Watson-Crick partners in the first half of the draws: hydrogen bonds need them
    pairs = torch.as_tensor(np.stack([np.arange(0, 2 * m, 2), np.arange(1, 2 * m, 2)], 1))
    none = torch.zeros((0, 2), dtype=torch.long)
    if model == 4:
        bt, ut = orc.pair_terms_na1(P, torch.as_tensor(c), torch.as_tensor(q), torch.as_tensor(seq), torch.as_tensor(is_rna),
                                    torch.ones(2 * m, dtype=torch.long), pairs if bonded else none, none if bonded else pairs)
    else:
        bt, ut = orc.pair_terms(model, P, torch.as_tensor(c), torch.as_tensor(q), torch.as_tensor(seq), torch.ones(2 * m, dtype=torch.long),
                                pairs if bonded else none, none if bonded else pairs)
    terms = {k: v.numpy() for k, v in (bt if bonded else ut).items()}
    radial = terms["bonded_excluded_volume" if bonded else "unbonded_excluded_volume"]
    angular = ("stacking",) if bonded else ("hydrogen_bonding", "cross_stacking", "coaxial_stacking")
    ok = np.isfinite(sum(terms.values())) & (radial < 20.0)
    chosen = []
    for k in angular:  # an equal share for every angular term, the most strongly live first
        idx = np.nonzero(ok & (np.abs(terms[k]) > 1e-3))[0]
        chosen.append(idx[np.argsort(-np.abs(terms[k][idx]))][:: max(1, len(idx) // (keep // len(angular)))][: keep // len(angular)])
    sel = np.unique(np.concatenate(chosen))
    rest = np.setdiff1d(np.nonzero(ok)[0], sel)[: keep - len(sel)]
    sel = np.concatenate([sel, rest])
    live = {k: int((np.abs(terms[k][sel]) > 1e-3).sum()) for k in terms}
    side = int(np.ceil(len(sel) ** (1.0 / 3.0)))
    grid = np.array([[i % side, (i // side) % side, i // (side * side)] for i in range(len(sel))], dtype=np.float64) * 8.0
    nuc = np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1)
    center = c[nuc] + np.repeat(grid, 2, axis=0)
    top = topology.from_arrays(seq[nuc].astype(np.int32), [2] * len(sel) if bonded else [1] * (2 * len(sel)))
    if model == 4:
        import dataclasses

        nt = np.where(is_rna[nuc], int(topology.NucleotideType.RNA), int(topology.NucleotideType.DNA)).astype(np.int32)
        top = dataclasses.replace(top, nt_type=nt)
        kinds = is_rna[nuc].reshape(-1, 2)
        live["pairs dna-dna / rna-rna / hybrid"] = (int((~kinds).all(1).sum()), int(kinds.all(1).sum()), int((kinds[:, 0] != kinds[:, 1]).sum()))
    return top, np.ascontiguousarray(center), np.ascontiguousarray(q[nuc]), live


# ---- the known answers the reference's own observable tests hold (mythos/observables/tests/test_rise.py:14-83,
#      test_propeller.py:13-80), as inputs our classes take.  The reference's mocks make the base site / base normal of a
#      nucleotide its CENTRE; here: site offsets 0 (base site = centre) and quaternions whose a3 is the wanted normal.
#      test_rise_call indexes site 3 of a 3-site frame; JAX clamps out-of-range gathers, so its fourth site IS the third.
#      (test_lp.py:50-59 needs data/test-data/simple-helix-60bp/output.dat, which this snapshot of the reference does not hold.)
ZERO_GEOMETRY = {"com_to_hb": 0.0, "com_to_stacking": 0.0, "com_to_backbone_x": 0.0, "com_to_backbone_y": 0.0}
RISE_SINGLE = {"quartets": [[[0, 1], [1, 2]]], "centers": [[0, 0, 0], [1, 1, 1], [2, 2, 2], [3, 3, 3]], "expected": 14.753608}
RISE_CALL = {"quartets": [[[0, 1], [1, 2]], [[1, 2], [2, 3]]], "centers": [[0, 0, 0], [1, 1, 1], [2, 2, 2], [2, 2, 2]],
             "frames": 5, "expected": 11.065206}
_S = 0.5 ** 0.5
# a3 = e_x, e_y, e_z, e_x (TEST_NORMALS): z -> x is +90 degrees about y, z -> y is -90 degrees about x
PROPELLER_CALL = {"pairs": [[0, 1], [0, 2], [0, 3]], "quats": [[_S, 0, _S, 0], [_S, -_S, 0, 0], [1, 0, 0, 0], [_S, 0, _S, 0]],
                  "frames": 5, "expected": 120.0}


def load_regr(name: str):
    """oxDNA regression runs the reference ships (tests/golden/regr/<name>): oxDNA2, half-charged ends, average sequence,
    circular strands.  -> (topology, trajectory, oxDNA's split energies per nucleotide for the trajectory's frames (F, 8),
    bonded pairs with every ring's closing pair turned into strand direction (last, first))."""
    base = GOLDEN / "regr" / name
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        top = topology.from_oxdna_file(base / ("sys.top" if (base / "sys.top").exists() else "generated.top"))
    traj = trajectory.from_file(base / "output.dat", top.strand_counts, is_5p_3p=False)
    split = np.loadtxt(base / "split_energy.dat")  # rows: the start configuration, then one per printed frame
    n_frames = traj.center.shape[0]
    bonded = np.asarray(top.bonded_neighbors).copy()
    closing = np.nonzero(bonded[:, 1] - bonded[:, 0] != 1)[0]
    turned = bonded.copy()
    turned[closing] = turned[closing][:, ::-1]
    return top, traj, split[1 : n_frames + 1, 1:9], turned


def load_lammps_regr(name: str):
    """LAMMPS oxDNA2 runs the reference ships (tests/golden/regr/lammps-oxdna2-40bp[-sa]): -> (topology, trajectory of the
    dumped steps (TacoxDNA conversion), per-nucleotide energies of those steps from LAMMPS's log as a dict of arrays:
    bond (FENE), hb, excv (NON-bonded pairs only: LAMMPS's pair styles skip bonded neighbours), stk, xstk, coax, dh)."""
    base = GOLDEN / "regr" / name
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        top = topology.from_oxdna_file(base / "data.top")
    traj = trajectory.from_file(base / "data.oxdna", top.strand_counts, is_5p_3p=False)
    rows, on = [], False
    for ln in (base / "log.lammps").read_text().splitlines():
        if ln.startswith("v_tns Temp"):  # thermo_style custom v_tns temp evdwl ecoul ebond eangle edihed pe v_cpuh c_hbond c_excv c_stk c_xstk c_coaxstk c_dh
            on = True
            continue
        if on:
            parts = ln.split()
            try:
                vals = [float(x) for x in parts]
            except ValueError:
                vals = []
            if len(vals) == 15:
                rows.append(vals)
            else:
                on = False
    rows = np.array(rows)[: traj.center.shape[0]]
    cols = {"bond": 4, "hb": 9, "excv": 10, "stk": 11, "xstk": 12, "coax": 13, "dh": 14}
    return top, traj, {k: rows[:, c] for k, c in cols.items()}
