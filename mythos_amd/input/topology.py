"""oxDNA topology files -> index arrays for the HIP force field.

Mirrors the behaviour of the reference loader (mythos/input/topology.py:123-327):
nucleotides are kept in oxDNA-classic 3'->5' memory order, a bonded pair is
``(k, k+1)`` inside one strand, a circular strand contributes ``(first, last)``
(:166-183), and the unbonded pair set is every ``i < j`` that is not bonded
(:186-190).  Differences that matter on a 288 GB GPU: the O(N^2) unbonded set is
materialised lazily (a 12 kbp duplex has 2.9e8 of them; the HIP path consumes a
cell-list instead) and all index arrays are contiguous int32, the width the
kernels read.
"""

from __future__ import annotations

import dataclasses as dc
import warnings
from enum import IntEnum
from pathlib import Path

import numpy as np

NUCLEOTIDES_IDX = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3}

ERR_INVALID_NUMBER_NUCLEOTIDES = "Invalid number of nucleotides"
ERR_INVALID_STRAND_COUNTS = "Invalid strand counts"
ERR_STRAND_COUNTS_NOT_MATCH = "Strand counts do not match number of nucleotides"
ERR_BONDED_SHAPE = "Invalid bonded neighbors shape"
ERR_SEQ_SHAPE = "Invalid discrete sequence shape"
ERR_SEQ_NUCLEOTIDES = "Invalid sequence nucleotides"
ERR_INVALID_OXDNA_FORMAT = "Invalid oxDNA topology format (first line must have 2 or 3 tokens)"
ERR_FILE_NOT_FOUND = "Topology file not found"
ERR_CIRCULAR_MISMATCH = "Strand counts and circularity do not match"
ERR_UNBONDED_TOO_LARGE = (
    "Refusing to materialise {n} all-pairs unbonded neighbours; use a neighbor list "
    "(mythos_amd.simulators.neighbors.CellNeighborList) for systems this large"
)

# all-pairs lists above this many nucleotides are refused (8000 -> 3.2e7 pairs, 256 MB int32)
MAX_ALL_PAIRS_N = 8192


class NucleotideType(IntEnum):
    UNSPECIFIED = 0
    DNA = 1
    RNA = 2


class oxDNAFormat(IntEnum):  # noqa: N801 - reference spelling (mythos/utils/types.py)
    CLASSIC = 0
    NEW = 1


def bonded_pairs(strand_lengths, is_circular) -> np.ndarray:
    """(B,2) int32 bonded pairs, reference ordering (mythos/input/topology.py:166-183)."""
    if len(strand_lengths) != len(is_circular):
        raise ValueError(ERR_CIRCULAR_MISMATCH)
    out = []
    start = 0
    for length, circ in zip(strand_lengths, is_circular):
        length = int(length)
        idx = np.arange(start, start + length - 1, dtype=np.int32)
        out.append(np.stack([idx, idx + 1], axis=1))
        if circ:
            out.append(np.array([[start, start + length - 1]], dtype=np.int32))
        start += length
    if not out:
        return np.zeros((0, 2), dtype=np.int32)
    return np.concatenate(out, axis=0).astype(np.int32)


def unbonded_pairs(n: int, bonded: np.ndarray) -> np.ndarray:
    """All i<j pairs that are not bonded, sorted lexicographically, (P,2) int32."""
    if n > MAX_ALL_PAIRS_N:
        raise MemoryError(ERR_UNBONDED_TOO_LARGE.format(n=n * (n - 1) // 2))
    iu, ju = np.triu_indices(n, k=1)
    keep = np.ones(iu.shape[0], dtype=bool)
    if len(bonded):
        lo = np.minimum(bonded[:, 0], bonded[:, 1]).astype(np.int64)
        hi = np.maximum(bonded[:, 0], bonded[:, 1]).astype(np.int64)
        # position of (lo,hi) in the row-major upper triangle
        pos = lo * n - lo * (lo + 1) // 2 + (hi - lo - 1)
        keep[pos] = False
    return np.stack([iu[keep], ju[keep]], axis=1).astype(np.int32)


@dc.dataclass(frozen=True)
class Topology:
    """Index arrays of one oxDNA system (reference: mythos/input/topology.py:85-120)."""

    n_nucleotides: int
    strand_counts: np.ndarray
    bonded_neighbors: np.ndarray
    seq: np.ndarray
    is_end: np.ndarray
    nt_type: np.ndarray
    is_circular: tuple = ()
    _unbonded: np.ndarray | None = dc.field(default=None, repr=False, compare=False)

    def __post_init__(self) -> None:
        if self.n_nucleotides < 1:
            raise ValueError(ERR_INVALID_NUMBER_NUCLEOTIDES)
        if len(self.strand_counts) == 0 or int(np.sum(self.strand_counts)) == 0:
            raise ValueError(ERR_INVALID_STRAND_COUNTS)
        if self.n_nucleotides != int(np.sum(self.strand_counts)):
            raise ValueError(ERR_STRAND_COUNTS_NOT_MATCH)
        b = np.asarray(self.bonded_neighbors)
        if b.ndim != 2 or b.shape[1] != 2:
            raise ValueError(ERR_BONDED_SHAPE)
        s = np.asarray(self.seq)
        if s.shape != (self.n_nucleotides,):
            raise ValueError(ERR_SEQ_SHAPE)
        if len(set(np.unique(s).tolist()) - {0, 1, 2, 3}) > 0:
            raise ValueError(ERR_SEQ_NUCLEOTIDES)

    @property
    def unbonded_neighbors(self) -> np.ndarray:
        """(P,2) int32, every i<j unbonded pair (reference semantics; O(N^2))."""
        if self._unbonded is None:
            object.__setattr__(self, "_unbonded", unbonded_pairs(self.n_nucleotides, self.bonded_neighbors))
        return self._unbonded

    @property
    def bonded_partners(self) -> np.ndarray:
        """(N,4) int32, the slot convention of the kernels' neighbour rows (mythos_amd/csrc/mythos_internal.h): slot 0 =
        partner of which this nucleotide is ``nn_j`` (bond (partner, self)), slot 1 = partner of which it is ``nn_i``
        (bond (self, partner)), slots 2 / 3 = a second partner in the same role, -1 where there is none.  Only the two
        ends of a circular strand use slots 2 / 3: the ring is closed by the pair (first, last) in that order
        (reference: input/topology.py:178-180), so ``first`` is nn_i of two bonds and ``last`` is nn_j of two."""
        part = np.full((self.n_nucleotides, 4), -1, dtype=np.int32)
        for i, j in np.asarray(self.bonded_neighbors):
            # bond (i, j): i is the "nn_i" role (odd slots), j the "nn_j" role (even slots)
            part[i, 1 if part[i, 1] < 0 else 3] = j
            part[j, 0 if part[j, 0] < 0 else 2] = i
        return part


def from_arrays(seq: np.ndarray, strand_counts, is_circular=None) -> Topology:
    """Build a topology for synthetic systems (benchmarks, generators)."""
    strand_counts = np.asarray(strand_counts, dtype=np.int64)
    if is_circular is None:
        is_circular = [False] * len(strand_counts)
    is_end = np.zeros(int(strand_counts.sum()), dtype=np.int32)
    start = 0
    for length, circ in zip(strand_counts, is_circular):
        if not circ:
            is_end[start] = 1
            is_end[start + int(length) - 1] = 1
        start += int(length)
    n = int(strand_counts.sum())
    return Topology(
        n_nucleotides=n,
        strand_counts=strand_counts,
        bonded_neighbors=bonded_pairs(strand_counts, is_circular),
        seq=np.asarray(seq, dtype=np.int32),
        is_end=is_end,
        nt_type=np.full(n, int(NucleotideType.DNA), dtype=np.int32),
        is_circular=tuple(bool(c) for c in is_circular),
    )


def _parse_classic(lines: list[str]) -> Topology:
    n_nucleotides, n_strands = (int(t) for t in lines[0].split())
    rows = [ln.split() for ln in lines[1:] if ln.strip()]
    strand_ids = np.array([int(r[0]) for r in rows])
    bases = [r[1] for r in rows]
    nbr_5p = [int(r[3]) for r in rows]
    strand_counts = np.array([int(np.sum(strand_ids == s)) for s in range(1, n_strands + 1)])
    is_circular, is_end, nt_type = [], [], []
    for s in range(1, n_strands + 1):
        members = np.nonzero(strand_ids == s)[0]
        sb = [bases[k] for k in members]
        circ = nbr_5p[members[-1]] != -1
        is_circular.append(circ)
        ends = [0] * len(sb)
        if not circ:
            ends[0] = 1
            ends[-1] = 1
        is_end.extend(ends)
        if "T" in sb:
            nt_type.extend([NucleotideType.DNA] * len(sb))
        elif "U" in sb:
            nt_type.extend([NucleotideType.RNA] * len(sb))
        else:
            warnings.warn(f"Type of strand {s} not specified, and did not find T/U for autodetect", stacklevel=1)
            nt_type.extend([NucleotideType.UNSPECIFIED] * len(sb))
    # bases are listed strand by strand in file order already
    order = np.concatenate([np.nonzero(strand_ids == s)[0] for s in range(1, n_strands + 1)])
    seq = np.array([NUCLEOTIDES_IDX[bases[k]] for k in order], dtype=np.int32)
    return Topology(
        n_nucleotides=n_nucleotides,
        strand_counts=strand_counts,
        bonded_neighbors=bonded_pairs(strand_counts, is_circular),
        seq=seq,
        is_end=np.array(is_end, dtype=np.int32),
        nt_type=np.array(nt_type, dtype=np.int32),
        is_circular=tuple(is_circular),
    )


def _parse_new(lines: list[str]) -> Topology:
    n_nucleotides = int(lines[0].split()[0])
    seq, strand_counts, is_circular, is_end, nt_type = [], [], [], [], []
    for ln in lines[1:]:
        if not ln.strip():
            continue
        nts = ln.split()[0]
        seq.append(nts[::-1])  # file is 5'->3'; memory order is 3'->5'
        strand_counts.append(len(nts))
        circ = "circular=true" in ln
        is_circular.append(circ)
        ends = [0] * len(nts)
        if not circ:
            ends[0] = 1
            ends[-1] = 1
        is_end.extend(ends)
        if "type=DNA" in ln:
            nt_type.extend([NucleotideType.DNA] * len(nts))
        elif "type=RNA" in ln:
            nt_type.extend([NucleotideType.RNA] * len(nts))
        else:
            warnings.warn(f"Type of strand {ln.strip()} not specified", stacklevel=1)
            nt_type.extend([NucleotideType.UNSPECIFIED] * len(nts))
    seq = "".join(seq)
    return Topology(
        n_nucleotides=n_nucleotides,
        strand_counts=np.array(strand_counts),
        bonded_neighbors=bonded_pairs(strand_counts, is_circular),
        seq=np.array([NUCLEOTIDES_IDX[c] for c in seq], dtype=np.int32),
        is_end=np.array(is_end, dtype=np.int32),
        nt_type=np.array(nt_type, dtype=np.int32),
        is_circular=tuple(is_circular),
    )


def from_oxdna_file(path, *, return_format: bool = False):
    """Read an oxDNA topology (classic or new format)."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(ERR_FILE_NOT_FOUND)
    lines = path.read_text().splitlines()
    n_tok = len(lines[0].split())
    if n_tok == 2:
        fmt, top = oxDNAFormat.CLASSIC, _parse_classic(lines)
    elif n_tok == 3:
        fmt, top = oxDNAFormat.NEW, _parse_new(lines)
    else:
        raise ValueError(ERR_INVALID_OXDNA_FORMAT)
    return (top, fmt) if return_format else top
