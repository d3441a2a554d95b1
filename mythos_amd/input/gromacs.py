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

    @classmethod
    def from_tpr(cls, path) -> "MartiniTopology":
        """Names, residues, bonds and angles from a GROMACS run input file (the reference goes through
        MDAnalysis.Universe(tpr), mythos/energy/martini/base.py:76-84)."""
        t = read_tpr_topology(path)
        return cls(atom_types=t["atom_types"], atom_names=t["atom_names"], residue_names=t["residue_names"],
                   angles=t["angles"], bonded_neighbors=t["bonds"])

    def tile(self, reps: int) -> "MartiniTopology":
        """The same molecules repeated ``reps`` times (for tiled boxes)."""
        n = len(self.atom_types)
        return MartiniTopology(
            atom_types=self.atom_types * reps, atom_names=self.atom_names * reps, residue_names=self.residue_names * reps,
            angles=np.concatenate([self.angles + k * n for k in range(reps)]),
            bonded_neighbors=np.concatenate([self.bonded_neighbors + k * n for k in range(reps)]),
        )


# ---- .tpr -------------------------------------------------------------------------------------------------------------
# Function types (GROMACS ifunc.h order) whose parameter records this reader knows, with their number of reals; a
# topology with any other type is refused rather than mis-read.
_TPR_IPARAM_REALS = {0: 4, 1: 4, 5: 4, 10: 4, 11: 4, 16: 6, 33: 4, 37: 2, 62: 2, 63: 2}
_F_BONDS, _F_G96BONDS, _F_HARMONIC, _F_ANGLES, _F_G96ANGLES = 0, 1, 5, 10, 11
_TPR_F_NRE = {137: 95}  # interaction lists per molecule type, by tpx version


class _Xdr:
    """Big-endian cursor over the in-memory serialisation GROMACS uses for the tpr body (tpx generation >= 27):
    fixed-size fields at their natural width, strings as int64 length + bytes."""

    def __init__(self, data: bytes, off: int = 0):
        self.d, self.o = data, off

    def _take(self, fmt: str, size: int):
        v = struct.unpack_from(fmt, self.d, self.o)[0]
        self.o += size
        return v

    def i4(self):
        return self._take(">i", 4)

    def u2(self):
        return self._take(">H", 2)

    def u1(self):
        return self._take(">B", 1)

    def i8(self):
        return self._take(">q", 8)

    def f4(self):
        return self._take(">f", 4)

    def f8(self):
        return self._take(">d", 8)

    def ints(self, n: int) -> list:
        v = list(struct.unpack_from(f">{n}i", self.d, self.o))
        self.o += 4 * n
        return v

    def xdr_string(self) -> str:  # header strings: XDR (size incl. terminator, length, bytes padded to 4)
        self.i4()
        n = self.i4()
        v = self.d[self.o:self.o + n].decode("latin1")
        self.o += (n + 3) // 4 * 4
        return v

    def string(self) -> str:
        n = self.i8()
        v = self.d[self.o:self.o + n].decode("latin1")
        self.o += n
        return v


def read_tpr_topology(path) -> dict:
    """Atom types / names / residue names and the bond and angle index lists of a GROMACS ``.tpr``.

    A narrow reader, written from the layout of GROMACS' tpxio serialisation: single precision, tpx version 137
    (GROMACS 2025), force fields made of bonds, (G96) angles, constraints and LJ tables - what MARTINI lipid systems
    use.  Anything else raises ``ValueError`` instead of guessing.  Walks: header, box, symbol table, force-field
    parameter table, molecule types (atoms, interaction lists, exclusions) and molecule blocks; the result is
    checked against the atom count in the header."""
    data = Path(path).read_bytes()
    r = _Xdr(data)
    version = r.xdr_string()
    precision, fver, fgen = r.i4(), r.i4(), r.i4()
    if not version.startswith("VERSION") or precision != 4 or fgen < 27 or fver not in _TPR_F_NRE:
        raise ValueError(f"{path}: unsupported tpr ({version!r}, precision {precision}, tpx version {fver}, generation {fgen})")
    r.xdr_string()  # file tag
    natoms, ngtc = r.i4(), r.i4()
    r.i4(), r.f4()  # fep state, lambda
    has_ir, has_top, has_x, has_v, has_f, has_box = (r.i4() for _ in range(6))
    body = r.i8()
    if not has_top or r.o + body != len(data):
        raise ValueError(f"{path}: no topology in this tpr, or a truncated file")
    if has_box:
        r.o += 3 * 9 * 4  # box, relative box, box velocity
    r.o += 4 * ngtc       # thermostat integrals of old file versions (kept for alignment)
    syms = [r.string() for _ in range(r.i4())]
    r.i4()                # system name
    r.i4()                # number of atom types
    ftypes = r.ints(r.i4())
    r.f8(), r.f4()        # reppow, fudgeQQ
    for ft in ftypes:
        if ft not in _TPR_IPARAM_REALS:
            raise ValueError(f"{path}: interaction function type {ft} is not known to this reader")
        r.o += 4 * _TPR_IPARAM_REALS[ft]
    f_nre = _TPR_F_NRE[fver]
    moltypes = []
    for _ in range(r.i4()):
        r.i4()  # molecule type name
        nat, nres = r.i4(), r.i4()
        resind = []
        for _ in range(nat):
            r.o += 16      # m, q, mB, qB
            r.u2(), r.u2()  # type, typeB
            r.i4()         # particle type
            resind.append(r.i4())
            r.i4()         # atomic number
        names = [syms[k] for k in r.ints(nat)]
        types = [syms[k] for k in r.ints(nat)]
        r.ints(nat)        # type names of the B state
        resnames = []
        for _ in range(nres):
            resnames.append(syms[r.i4()])
            r.i4(), r.u1()  # residue number, insertion code
        ilists = [r.ints(r.i4()) for _ in range(f_nre)]
        r.ints(r.i4() + 1)  # the obsolete charge-group index
        n_excl, n_excl_a = r.i4(), r.i4()
        r.ints(n_excl + 1 + n_excl_a)
        bonds = [tuple(il[k + 1:k + 3]) for ft in (_F_BONDS, _F_G96BONDS, _F_HARMONIC) for il in [ilists[ft]] for k in range(0, len(il), 3)]
        angles = [tuple(il[k + 1:k + 4]) for ft in (_F_ANGLES, _F_G96ANGLES) for il in [ilists[ft]] for k in range(0, len(il), 4)]
        moltypes.append({"names": names, "types": types, "res": [resnames[k] for k in resind], "bonds": bonds, "angles": angles})
    out = {"atom_types": [], "atom_names": [], "residue_names": [], "bonds": [], "angles": []}
    base = 0
    for _ in range(r.i4()):
        mt, nmol, nat_mol = moltypes[r.i4()], r.i4(), r.i4()
        r.i4(), r.i4()  # position-restraint counts
        if nat_mol != len(mt["names"]):
            raise ValueError(f"{path}: molecule block disagrees with its molecule type")
        for _ in range(nmol):
            out["atom_types"] += mt["types"]
            out["atom_names"] += mt["names"]
            out["residue_names"] += mt["res"]
            out["bonds"] += [(base + i, base + j) for i, j in mt["bonds"]]
            out["angles"] += [(base + i, base + j, base + k) for i, j, k in mt["angles"]]
            base += nat_mol
    if base != natoms or r.i4() != natoms:
        raise ValueError(f"{path}: {base} atoms in the molecule blocks, {natoms} in the header")
    return {"atom_types": tuple(out["atom_types"]), "atom_names": tuple(out["atom_names"]), "residue_names": tuple(out["residue_names"]),
            "bonds": np.array(out["bonds"], dtype=np.int32).reshape(-1, 2), "angles": np.array(out["angles"], dtype=np.int32).reshape(-1, 3)}


def read_xvg(path) -> np.ndarray:
    """Second column of a ``gmx energy`` .xvg file."""
    vals = []
    for ln in Path(path).read_text().splitlines():
        if ln and not ln.startswith(("#", "@")):
            vals.append(float(ln.split()[1]))
    return np.array(vals)
