"""Shared pieces of the structural observables (mythos/observables/base.py:13-66): the quartets of a duplex, and the
HIP evaluation every observable class goes through.

An observable object describes WHAT to measure (index lists, geometry, displacement); the numbers come from one
workgroup per frame on the GPU (mythos_amd/csrc/observables.h) - either a stand-alone launch, or, when the energy
function was built ``with_observables(...)``, the same library call that evaluates the energies and
dU/dtheta of the frames (``mythos_oxdna_energy_obs``), so a DiffTRe iteration reads its trajectory once.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from mythos_amd import _lib

ANGSTROMS_PER_OXDNA_LENGTH = 8.518  # mythos/utils/units.py:5-8

# output row of mythos_observables_eval / mythos_oxdna_energy_obs
COL_PROPELLER, COL_RISE, COL_PITCH, COL_L0, COL_CORR = 0, 1, 2, 3, 4


def get_duplex_quartets(n_nucs_per_strand: int) -> torch.Tensor:
    """All pairs of adjacent base pairs of a duplex whose strands are stored one after the other:
    base pair k = (k, 2n - 1 - k); quartet k = (base pair k, base pair k + 1).  Shape (n - 1, 2, 2)
    (mythos/observables/base.py:48-66)."""
    n = int(n_nucs_per_strand)
    k = torch.arange(n)
    bps = torch.stack([k, 2 * n - 1 - k], dim=1)
    return torch.stack([bps[:-1], bps[1:]], dim=1)


def _geometry3(geometry: dict, model: int) -> np.ndarray:
    if model == 3:  # oxRNA2: the third entry is the backbone offset along a3 (rna2/nucleotide.py:56)
        return np.array([float(geometry["pos_base"]), float(geometry["pos_back_a1"]), float(geometry["pos_back_a3"])])
    if model == 2:
        return np.array([float(geometry["com_to_hb"]), float(geometry["com_to_backbone_x"]), float(geometry["com_to_backbone_y"])])
    return np.array([float(geometry["com_to_hb"]), float(geometry["com_to_backbone"]), 0.0])


class ObservableSet:
    """mythos_obs_t: the index lists of up to one propeller-twist list and one quartet list on the device."""

    def __init__(self, n: int, model: int, geometry: dict | None, box, base_pairs, quartets, skip_ends: bool, dtype, device):
        lib = _lib.load()
        self.n, self.model, self.dtype, self.device = int(n), int(model), dtype, torch.device(device)
        bps = np.ascontiguousarray(np.asarray(base_pairs if base_pairs is not None else np.zeros((0, 2)), dtype=np.int32).reshape(-1, 2))
        qs = np.ascontiguousarray(np.asarray(quartets if quartets is not None else np.zeros((0, 2, 2)), dtype=np.int32).reshape(-1, 2, 2))
        if qs.shape[0] > 0 and geometry is None:
            raise ValueError("rise, pitch and persistence length need the [geometry] section (site offsets)")
        g3 = _geometry3(geometry, model) if geometry is not None else np.zeros(3)
        box_arr = None if box is None else np.ascontiguousarray(np.broadcast_to(np.asarray(box, np.float64), (3,)))
        self._h = lib.mythos_observables_create(
            self.model, self.n, g3.ctypes.data_as(_lib.c_double_p), None if box_arr is None else box_arr.ctypes.data_as(_lib.c_double_p),
            int(bps.shape[0]), bps.ctypes.data_as(_lib.c_int_p), int(qs.shape[0]), qs.ctypes.data_as(_lib.c_int_p), int(bool(skip_ends)),
            0 if dtype == torch.float32 else 1, self.device.index or 0)
        if not self._h:
            raise _lib.MythosHipError(f"mythos_observables_create: {_lib.last_error()}")
        self._lib = lib
        self.width = int(lib.mythos_observables_width(self._h))
        self.n_corr = self.width - COL_CORR

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mythos_observables_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def eval(self, center: torch.Tensor, quat: torch.Tensor) -> torch.Tensor:
        """(S, width) float64 rows for (S, n, 3) / (S, n, 4) frames - the stand-alone launch."""
        c, q = center.contiguous(), quat.contiguous()
        out = torch.empty((c.shape[0], self.width), dtype=torch.float64, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib.mythos_observables_eval(self._h, _lib.ptr(c), _lib.ptr(q), int(c.shape[0]), _lib.ptr(out), stream),
                   "observables_eval")
        return out


def _frames(trajectory):
    c, q = trajectory.center, trajectory.orientation.vec
    if c.dim() == 2:
        c, q = c[None], q[None]
    if c.device.type != "cuda":
        raise _lib.MythosHipError("observables are evaluated by the HIP library: the trajectory must live on a GPU "
                                  "(mythos_amd has no CPU fallback; oracle/observables_oracle.py is the test-side checker)")
    if c.dtype not in (torch.float32, torch.float64):
        raise ValueError(f"unsupported dtype {c.dtype}")
    return c, q.to(c.dtype)


# Rows computed alongside an energy launch (mythos_oxdna_energy_obs), remembered together with the frames they belong to.  An entry
# HOLDS its two tensors: while it lives their memory cannot go back to the allocator and be handed to another trajectory
# of the same shape, so "same address, same shape, same version counters" does identify the frames (an entry keyed on
# data_ptr alone would serve the rows of a freed trajectory to the next one allocated in its place).  The library
# writes into caller tensors through raw pointers in two places only - LangevinIntegrator.run / store - and those bump
# the version counters (hip_system._touched).  At most four entries, i.e. at most four trajectories kept alive here - and
# only until the optimisation step that produced them ends: SimpleOptimizer.step and DiffTReObjective.calculate call
# clear_fused() when they are done with the trajectory (nothing did, and a DiffTRe loop over large
# trajectories kept four stale ones on the GPU).
_FUSED: list = []
_FUSED_MAX = 4


def _same_frames(entry, c: torch.Tensor, q: torch.Tensor) -> bool:
    ec, eq, cv, qv = entry[0], entry[1], entry[2], entry[3]
    return (ec.data_ptr() == c.data_ptr() and eq.data_ptr() == q.data_ptr() and tuple(ec.shape) == tuple(c.shape)
            and ec.dtype == c.dtype and ec._version == cv == c._version and eq._version == qv == q._version)


def remember_fused(c, q, signature, rows) -> None:
    _FUSED[:] = [e for e in _FUSED if not (_same_frames(e, c, q) and e[4] == signature)]
    while len(_FUSED) >= _FUSED_MAX:
        _FUSED.pop(0)
    _FUSED.append((c, q, c._version, q._version, signature, rows))


def lookup_fused(c, q, signature):
    for e in reversed(_FUSED):
        if e[4] == signature and _same_frames(e, c, q):
            return e[5]
    return None


def clear_fused() -> None:
    _FUSED.clear()


class HipObservable:
    """What the four observable classes share: the description of their index lists and the row lookup."""

    base_pairs = None
    quartets = None
    skip_ends = True
    geometry: dict | None = None
    model: int = 2
    displacement_fn = None

    def signature(self):
        bp = None if self.base_pairs is None else np.asarray(self.base_pairs, dtype=np.int32).tobytes()
        qs = None if self.quartets is None else np.asarray(self.quartets, dtype=np.int32).tobytes()
        box = getattr(self.displacement_fn, "box", None)
        box = None if box is None else tuple(np.broadcast_to(np.asarray(box, np.float64), (3,)).tolist())
        geo = None if self.geometry is None else tuple(_geometry3(self.geometry, self.model).tolist())
        return (bp, qs, bool(self.skip_ends), geo, self.model, box)

    def make_set(self, n: int, dtype, device) -> ObservableSet:
        box = getattr(self.displacement_fn, "box", None)
        return ObservableSet(n, self.model, self.geometry, box, self.base_pairs, self.quartets, self.skip_ends, dtype, device)

    def rows(self, trajectory) -> torch.Tensor:
        """(S, width) rows of this observable's set for the frames: from the energy launch that already produced
        them, or from a stand-alone launch."""
        c, q = _frames(trajectory)
        hit = lookup_fused(c, q, self.signature())
        if hit is not None:
            return hit
        cache = self.__dict__.setdefault("_sets", {})
        key = (int(c.shape[1]), c.dtype, str(c.device))
        if key not in cache:
            cache[key] = self.make_set(int(c.shape[1]), c.dtype, c.device)
        return cache[key].eval(c, q)
