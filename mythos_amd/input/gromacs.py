"""Minimal GROMACS readers for MARTINI systems: ``.trr`` (XDR), ``.gro`` and the
``[moleculetype]`` / ``[molecules]`` part of a ``.top``.

The reference reads these through MDAnalysis from a binary ``.tpr``
(mythos/simulators/gromacs/utils.py:20-60, mythos/energy/martini/base.py:46-94); MDAnalysis is not
available here, so topology comes from the text ``topol.top`` the reference ships next to the
``.tpr`` and frames from the fixed-record ``.trr``.  Units stay GROMACS-native (nm, kJ/mol): the
reference's Angstrom x 0.1 round trip through MDAnalysis cancels.
"""

from __future__ import annotations

import dataclasses as dc
import re
import struct
from pathlib import Path

import numpy as np

TRR_MAGIC = 1993


def read_trr(path, skip_first: bool = True) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> (positions (S,N,3), box (S,3), time (S,)) in nm / ps; velocities and forces are skipped.

    ``skip_first`` drops the initial state exactly as the reference's reader does
    (mythos/simulators/gromacs/utils.py:40-42).
    """
    data = Path(path).read_bytes()
    off, xs, boxes, times = 0, [], [], []
    while off < len(data):
        magic, slen = struct.unpack_from(">ii", data, off)
        if magic != TRR_MAGIC:
            raise ValueError(f"bad TRR magic {magic} at byte {off}")
        off += 8
        (n,) = struct.unpack_from(">i", data, off)
        off += 4 + ((n + 3) // 4) * 4
        ir, e, box_size, vir, pres, top, sym, x_size, v_size, f_size, natoms, _step, _nre = struct.unpack_from(">13i", data, off)
        off += 52
        real = 8 if (box_size == 72 or (natoms and x_size == natoms * 24)) else 4
        fmt = ">d" if real == 8 else ">f"
        t, _lam = struct.unpack_from(fmt[0] + fmt[1] * 2, data, off)
        off += 2 * real
        off += ir + e
        box = np.frombuffer(data, dtype=f">f{real}", count=9, offset=off).reshape(3, 3) if box_size else np.zeros((3, 3))
        off += box_size + vir + pres + top + sym
        x = np.frombuffer(data, dtype=f">f{real}", count=natoms * 3, offset=off).reshape(natoms, 3) if x_size else None
        off += x_size + v_size + f_size
        if x is not None:
            xs.append(x.astype(np.float64))
            boxes.append(np.diag(box).astype(np.float64))
            times.append(float(t))
    s = 1 if skip_first else 0
    return np.array(xs[s:]), np.array(boxes[s:]), np.array(times[s:])


def read_gro(path) -> tuple[list[str], list[str], np.ndarray, np.ndarray]:
    """-> (residue names, atom names, positions (N,3) nm, box (3,))."""
    lines = Path(path).read_text().splitlines()
    n = int(lines[1])
    res, names, pos = [], [], []
    for ln in lines[2 : 2 + n]:
        res.append(ln[5:10].strip())
        names.append(ln[10:15].strip())
        pos.append([float(ln[20:28]), float(ln[28:36]), float(ln[36:44])])
    box = np.array([float(x) for x in lines[2 + n].split()[:3]])
    return res, names, np.array(pos), box


@dc.dataclass(frozen=True)
class MartiniTopology:
    """Bead types, names, residues, bonds and angles of a MARTINI system
    (mythos/energy/martini/base.py:46-94)."""

    atom_types: tuple
    atom_names: tuple
    residue_names: tuple
    angles: np.ndarray  # (n_angles, 3)
    bonded_neighbors: np.ndarray  # (n_bonds, 2)

    @property
    def bond_names(self) -> tuple:
        """RESIDUE_BEAD1_BEAD2 per bond (base.py:20-29)."""
        return tuple(f"{self.residue_names[b[0]]}_{self.atom_names[b[0]]}_{self.atom_names[b[1]]}" for b in self.bonded_neighbors)

    @property
    def angle_names(self) -> tuple:
        """RESIDUE_BEAD1_BEAD2_BEAD3 per angle (base.py:32-42)."""
        return tuple(
            f"{self.residue_names[a[0]]}_{self.atom_names[a[0]]}_{self.atom_names[a[1]]}_{self.atom_names[a[2]]}" for a in self.angles
        )

    @classmethod
    def from_top(cls, path) -> "MartiniTopology":
        """Expand ``[moleculetype]`` blocks by ``[molecules]`` counts of a GROMACS .top file."""
        mols: dict[str, dict] = {}
        order: list[tuple[str, int]] = []
        section, cur = None, None
        for raw in Path(path).read_text().splitlines():
            ln = raw.split(";")[0].strip()
            if not ln or ln.startswith("#"):
                continue
            m = re.match(r"\[\s*(\w+)\s*\]", ln)
            if m:
                section = m.group(1).lower()
                continue
            tok = ln.split()
            if section == "moleculetype":
                cur = {"atoms": [], "bonds": [], "angles": []}
                mols[tok[0]] = cur
            elif section == "atoms" and cur is not None:
                cur["atoms"].append((tok[1], tok[3], tok[4]))  # type, residue, atom name
            elif section == "bonds" and cur is not None:
                cur["bonds"].append((int(tok[0]) - 1, int(tok[1]) - 1))
            elif section == "angles" and cur is not None:
                cur["angles"].append((int(tok[0]) - 1, int(tok[1]) - 1, int(tok[2]) - 1))
            elif section == "molecules":
                order.append((tok[0], int(tok[1])))
        types, names, res, bonds, angles = [], [], [], [], []
        base = 0
        for name, count in order:
            mol = mols[name]
            for _ in range(count):
                for t, r, a in mol["atoms"]:
                    types.append(t)
                    res.append(r)
                    names.append(a)
                bonds += [(base + i, base + j) for i, j in mol["bonds"]]
                angles += [(base + i, base + j, base + k) for i, j, k in mol["angles"]]
                base += len(mol["atoms"])
        return cls(
            atom_types=tuple(types), atom_names=tuple(names), residue_names=tuple(res),
            angles=np.array(angles, dtype=np.int32).reshape(-1, 3),
            bonded_neighbors=np.array(bonds, dtype=np.int32).reshape(-1, 2),
        )

    def tile(self, reps: int) -> "MartiniTopology":
        """The same molecules repeated ``reps`` times (for tiled boxes)."""
        n = len(self.atom_types)
        return MartiniTopology(
            atom_types=self.atom_types * reps, atom_names=self.atom_names * reps, residue_names=self.residue_names * reps,
            angles=np.concatenate([self.angles + k * n for k in range(reps)]),
            bonded_neighbors=np.concatenate([self.bonded_neighbors + k * n for k in range(reps)]),
        )


def read_xvg(path) -> np.ndarray:
    """Second column of a ``gmx energy`` .xvg file."""
    vals = []
    for ln in Path(path).read_text().splitlines():
        if ln and not ln.startswith(("#", "@")):
            vals.append(float(ln.split()[1]))
    return np.array(vals)
