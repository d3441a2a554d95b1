"""Sequence constraints and probabilistic sequences, with the semantics of the reference's
``mythos/input/sequence_constraints.py`` (SequenceConstraints :73-127, from_bps :130-181, dseq_to_pseq :184-216).

A probabilistic sequence is the pair ``(unpaired_pseq (n_unpaired, 4), bp_pseq (n_bp, 4))``: a distribution over
A, C, G, T for every unpaired nucleotide and over the base-pair types AT, TA, GC, CG (``BP_TYPES``,
mythos/utils/constants.py:13-20) for every constrained base pair.  The energy terms with sequence-dependent weights
(stacking, hydrogen bonding) then use the EXPECTED weight of each pair (mythos/energy/utils.py:45-132); on the GPU the
expectation is formed inside the energy kernel from per-nucleotide marginals (``kernel_tables`` below)."""

from __future__ import annotations

import dataclasses as dc

import numpy as np

DNA_ALPHA = "ACGT"
N_NT = 4
BP_TYPES = ("AT", "TA", "GC", "CG")
N_BP_TYPES = 4
N_NT_PER_BP = 2
BP_IDXS = np.array([[DNA_ALPHA.index(a), DNA_ALPHA.index(b)] for a, b in BP_TYPES], dtype=np.int32)
BP_IDX_MAP = {(int(a), int(b)): k for k, (a, b) in enumerate(BP_IDXS)}

ERR_SEQ_CONSTRAINTS_INVALID_NUMBER_NUCLEOTIDES = "Invalid number of nucleotides"
ERR_SEQ_CONSTRAINTS_INVALID_UNPAIRED_SHAPE = "Invalid shape for unpaired nucleotides"
ERR_INVALID_BP_SHAPE = "Invalid shape for base pairs"
ERR_SEQ_CONSTRAINTS_INVALID_IS_UNPAIRED_SHAPE = "Invalid shape for array specifying if unpaired"
ERR_SEQ_CONSTRAINTS_INVALID_UNPAIRED_MAPPER_SHAPE = "Invalid shape for unpaired nucleotide index mapper"
ERR_SEQ_CONSTRAINTS_INVALID_BP_MAPPER_SHAPE = "Invalid shape for base pair index mapper"
ERR_SEQ_CONSTRAINTS_MISMATCH_NUM_TYPES = (
    "Number of nucleotides should equal the number of unpaired base pairs plus the number of coupled base pairs"
)
ERR_SEQ_CONSTRAINTS_INVALID_COVER = "Unpaired and coupled nucleotides do not cover all nucleotides"
ERR_SEQ_CONSTRAINTS_IS_UNPAIRED_INVALID_VALUES = "Array specifying if unpaired contains invalid values, can only be one-hot"
ERR_SEQ_CONSTRAINTS_INVALID_IS_UNPAIRED = "Array specifying if is_unpaired disagrees with list of unpaired nucleotides"
ERR_SEQ_CONSTRAINTS_PAIRED_NT_MAPPED_TO_UNPAIRED = "Base paired nucleotides cannot be mapped to an unpaired nucleotide"
ERR_SEQ_CONSTRAINTS_INCOMPLETE_UNPAIRED_MAPPED_IDXS = (
    "Map of position indices to indices of unpaired nucleotides does not cover number of unpaired nucleotides"
)
ERR_SEQ_CONSTRAINTS_UNPAIRED_NT_MAPPED_TO_PAIRED = "Unpaired nucleotides cannot be mapped to a base paired nucleotide"
ERR_SEQ_CONSTRAINTS_INCOMPLETE_BP_MAPPED_IDXS = (
    "Map of position indices to indices of base paired nucleotides does not cover number of base paired nucleotides"
)
ERR_BP_ARR_CONTAINS_DUPLICATES = "Array specifying base paired indices cannot contain duplicates"
ERR_INVALID_BP_INDICES = "Base paired indices must be between 0 and n_nucleotides-1"
ERR_DSEQ_TO_PSEQ_INVALID_BP = "Invalid base pair encountered when converting discrete sequence to probabilistic sequence"


def _check_mappers(n_unpaired, n_bp, unpaired, idx_to_unpaired_idx, idx_to_bp_idx) -> None:
    is_up = set(int(u) for u in unpaired)
    seen_up, seen_bp = set(), set()
    for idx, m in enumerate(idx_to_unpaired_idx):
        if idx in is_up:
            seen_up.add(int(m))
        elif m != -1:
            raise ValueError(ERR_SEQ_CONSTRAINTS_PAIRED_NT_MAPPED_TO_UNPAIRED)
    if seen_up != set(range(n_unpaired)):
        raise ValueError(ERR_SEQ_CONSTRAINTS_INCOMPLETE_UNPAIRED_MAPPED_IDXS)
    for idx, (b, w) in enumerate(idx_to_bp_idx):
        if idx not in is_up:
            seen_bp.add((int(b), int(w)))
        elif b != -1 or w != -1:
            raise ValueError(ERR_SEQ_CONSTRAINTS_UNPAIRED_NT_MAPPED_TO_PAIRED)
    if seen_bp != {(b, w) for b in range(n_bp) for w in (0, 1)}:
        raise ValueError(ERR_SEQ_CONSTRAINTS_INCOMPLETE_BP_MAPPED_IDXS)


@dc.dataclass(frozen=True, eq=False)
class SequenceConstraints:
    """Which nucleotides are free and which are coupled in base pairs (sequence_constraints.py:73-127)."""

    n_nucleotides: int
    n_unpaired: int
    n_bp: int
    is_unpaired: np.ndarray          # (n,) 0 / 1
    unpaired: np.ndarray             # (n_unpaired,) nucleotide indices
    bps: np.ndarray                  # (n_bp, 2)
    idx_to_unpaired_idx: np.ndarray  # (n,) position among the unpaired, or -1
    idx_to_bp_idx: np.ndarray        # (n, 2) (base pair, position inside it), or (-1, -1)

    def __post_init__(self) -> None:
        for f in ("is_unpaired", "unpaired", "bps", "idx_to_unpaired_idx", "idx_to_bp_idx"):
            object.__setattr__(self, f, np.asarray(getattr(self, f)))
        if self.n_nucleotides < 1:
            raise ValueError(ERR_SEQ_CONSTRAINTS_INVALID_NUMBER_NUCLEOTIDES)
        if self.unpaired.shape != (self.n_unpaired,):
            raise ValueError(ERR_SEQ_CONSTRAINTS_INVALID_UNPAIRED_SHAPE)
        if self.bps.shape != (self.n_bp, 2):
            raise ValueError(ERR_INVALID_BP_SHAPE)
        if self.is_unpaired.shape != (self.n_nucleotides,):
            raise ValueError(ERR_SEQ_CONSTRAINTS_INVALID_IS_UNPAIRED_SHAPE)
        if self.idx_to_unpaired_idx.shape != (self.n_nucleotides,):
            raise ValueError(ERR_SEQ_CONSTRAINTS_INVALID_UNPAIRED_MAPPER_SHAPE)
        if self.idx_to_bp_idx.shape != (self.n_nucleotides, 2):
            raise ValueError(ERR_SEQ_CONSTRAINTS_INVALID_BP_MAPPER_SHAPE)
        if self.n_unpaired + 2 * self.n_bp != self.n_nucleotides:
            raise ValueError(ERR_SEQ_CONSTRAINTS_MISMATCH_NUM_TYPES)
        if set(np.concatenate([self.unpaired, self.bps.reshape(-1)]).astype(int).tolist()) != set(range(self.n_nucleotides)):
            raise ValueError(ERR_SEQ_CONSTRAINTS_INVALID_COVER)
        if not set(np.asarray(self.is_unpaired).astype(int).tolist()).issubset({0, 1}):
            raise ValueError(ERR_SEQ_CONSTRAINTS_IS_UNPAIRED_INVALID_VALUES)
        up = set(int(u) for u in self.unpaired)
        for idx, flag in enumerate(self.is_unpaired):
            if bool(flag) != (idx in up):
                raise ValueError(ERR_SEQ_CONSTRAINTS_INVALID_IS_UNPAIRED)
        _check_mappers(self.n_unpaired, self.n_bp, self.unpaired, self.idx_to_unpaired_idx, self.idx_to_bp_idx)


def from_bps(n_nucleotides: int, bps) -> SequenceConstraints:
    """Constraints from a list of base pairs; every other nucleotide is unpaired (sequence_constraints.py:130-181)."""
    bps = np.asarray(bps)
    if bps.ndim != 2 or bps.shape[1] != N_NT_PER_BP or N_NT_PER_BP * bps.shape[0] > n_nucleotides:
        raise ValueError(ERR_INVALID_BP_SHAPE)
    bps = bps.astype(np.int32)
    paired = bps.reshape(-1)
    if len(np.unique(paired)) < len(paired):
        raise ValueError(ERR_BP_ARR_CONTAINS_DUPLICATES)
    if not np.all((paired >= 0) & (paired < n_nucleotides)):
        raise ValueError(ERR_INVALID_BP_INDICES)
    unpaired = np.setdiff1d(np.arange(n_nucleotides), paired).astype(np.int32)
    idx_to_unpaired_idx = np.full((n_nucleotides,), -1, dtype=np.int32)
    idx_to_unpaired_idx[unpaired] = np.arange(unpaired.shape[0], dtype=np.int32)
    idx_to_bp_idx = np.full((n_nucleotides, 2), -1, dtype=np.int32)
    for b, (i, j) in enumerate(bps):
        idx_to_bp_idx[i] = (b, 0)
        idx_to_bp_idx[j] = (b, 1)
    is_unpaired = np.zeros(n_nucleotides, dtype=np.int32)
    is_unpaired[unpaired] = 1
    return SequenceConstraints(n_nucleotides=n_nucleotides, n_unpaired=int(unpaired.shape[0]), n_bp=int(bps.shape[0]),
                               is_unpaired=is_unpaired, unpaired=unpaired, bps=bps, idx_to_unpaired_idx=idx_to_unpaired_idx,
                               idx_to_bp_idx=idx_to_bp_idx)


def dseq_to_pseq(dseq, sc: SequenceConstraints):
    """One-hot probabilistic sequence of a discrete one (sequence_constraints.py:184-216); a base pair that is not
    AT / TA / GC / CG raises.  With no base pairs the second array is one row of zeros, as in the reference."""
    dseq = np.asarray(dseq)
    up = np.zeros((sc.n_unpaired, N_NT), dtype=np.float64)
    for k, idx in enumerate(sc.unpaired):
        up[k, int(dseq[idx])] = 1.0
    bp = np.zeros((max(sc.n_bp, 1), N_BP_TYPES), dtype=np.float64)
    for k, (i, j) in enumerate(sc.bps):
        key = (int(dseq[i]), int(dseq[j]))
        if key not in BP_IDX_MAP:
            raise ValueError(ERR_DSEQ_TO_PSEQ_INVALID_BP)
        bp[k, BP_IDX_MAP[key]] = 1.0
    return up, bp


def kernel_tables(pseq, sc: SequenceConstraints):
    """What mythos_oxdna_set_pseq takes: per-nucleotide marginal base probabilities (n, 4), the unit of every
    nucleotide (2 * base pair + position inside it, or -1 if unpaired) and the base-pair type probabilities (n_bp, 4).
    Two nucleotides of different units are independent, so their joint is the product of the marginals; the two
    members of one base pair are tied through its type (mythos/energy/utils.py:45-132, cases 1-4)."""
    up, bp = (np.asarray(a.detach().cpu().numpy() if hasattr(a, "detach") else a, dtype=np.float64) for a in pseq)
    if up.shape != (sc.n_unpaired, N_NT):
        raise ValueError(f"unpaired pseq must have shape ({sc.n_unpaired}, 4), got {up.shape}")
    if sc.n_bp > 0 and bp.shape != (sc.n_bp, N_BP_TYPES):
        raise ValueError(f"base-pair pseq must have shape ({sc.n_bp}, 4), got {bp.shape}")
    n = sc.n_nucleotides
    marg = np.zeros((n, N_NT), dtype=np.float64)
    unit = np.full((n,), -1, dtype=np.int32)
    for idx in range(n):
        if sc.is_unpaired[idx]:
            marg[idx] = up[sc.idx_to_unpaired_idx[idx]]
        else:
            b, w = (int(v) for v in sc.idx_to_bp_idx[idx])
            unit[idx] = 2 * b + w
            for t in range(N_BP_TYPES):
                marg[idx, BP_IDXS[t, w]] += bp[b, t]
    return marg, unit, np.ascontiguousarray(bp[: max(sc.n_bp, 1)] if sc.n_bp > 0 else np.zeros((1, 4)))


def kernel_tables_torch(pseq, sc: SequenceConstraints):
    """The (marginals, base-pair type probabilities) of ``kernel_tables`` as torch functions of the two pseq arrays, so
    that dU/d(marginals) and dU/d(type probabilities) from the kernel flow back to them by autograd: an unpaired
    nucleotide's marginal IS its row; a paired nucleotide's marginal sums its pair's type probabilities."""
    import torch

    up, bp = (torch.as_tensor(a, dtype=torch.float64) for a in pseq)
    n = sc.n_nucleotides
    sel_up = torch.zeros((n, max(sc.n_unpaired, 1)), dtype=torch.float64)
    sel_bp = torch.zeros((n, N_NT, max(sc.n_bp, 1), N_BP_TYPES), dtype=torch.float64)
    for idx in range(n):
        if sc.is_unpaired[idx]:
            sel_up[idx, int(sc.idx_to_unpaired_idx[idx])] = 1.0
        else:
            b, w = (int(v) for v in sc.idx_to_bp_idx[idx])
            for t in range(N_BP_TYPES):
                sel_bp[idx, int(BP_IDXS[t, w]), b, t] = 1.0
    marg = torch.zeros((n, N_NT), dtype=torch.float64)
    if sc.n_unpaired > 0:
        marg = marg + sel_up[:, : sc.n_unpaired] @ up
    if sc.n_bp > 0:
        marg = marg + torch.einsum("iabt,bt->ia", sel_bp[:, :, : sc.n_bp], bp[: sc.n_bp])
    bp_rows = bp[: sc.n_bp] if sc.n_bp > 0 else torch.zeros((1, N_BP_TYPES), dtype=torch.float64)
    return marg, bp_rows
