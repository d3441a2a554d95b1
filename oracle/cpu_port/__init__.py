"""ctypes loader of oracle/_build/libmythos_cpu_port.so  --  TEST INFRASTRUCTURE ONLY.

The C++/OpenMP port of the force evaluation and the Langevin step (oracle/cpu_port/md_cpu.cpp): bench.py's
``cpu_baseline`` leg and tests/test_cpu_port.py use it; nothing under mythos_amd/ may import this module.
Built by ``make -C oracle`` (``__graft_entry__.build()`` does that).
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

BUILD = Path(__file__).resolve().parent.parent / "_build"
LIB = BUILD / "libmythos_cpu_port.so"

_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB.exists():
            raise FileNotFoundError(f"{LIB} not built: run `make -C oracle`")
        lib = C.CDLL(str(LIB))
        V = C.c_void_p
        lib.mythos_cpu_create.restype = V
        lib.mythos_cpu_create.argtypes = [C.c_int, C.c_int, _ip, C.POINTER(C.c_uint8), C.c_int, _ip, _dp, _dp, C.c_int]
        lib.mythos_cpu_destroy.argtypes = [V]
        lib.mythos_cpu_set_types.argtypes = [V, C.POINTER(C.c_uint8)]
        lib.mythos_cpu_destroy.restype = None
        lib.mythos_cpu_threads.restype = C.c_int
        lib.mythos_cpu_set_threads.argtypes = [C.c_int]
        lib.mythos_cpu_set_pairs.argtypes = [V, _ip, C.c_int]
        lib.mythos_cpu_build_pairs.argtypes = [V, _dp, C.c_double]
        lib.mythos_cpu_mean_row.argtypes = [V]
        lib.mythos_cpu_mean_row.restype = C.c_double
        lib.mythos_cpu_energy.argtypes = [V, _dp, _dp, _dp, _dp, _dp, _dp]
        lib.mythos_cpu_langevin_run.argtypes = [V, _dp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                                C.c_double, _dp, C.c_uint64, C.c_int64, C.c_double, C.c_double, C.c_int, _dp]
        _lib = lib
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


class CpuPort:
    """One oxDNA system on the host.  ``flat``: the parameter vector of the C ABI (mythos_oxdna_param_name order)."""

    def __init__(self, model, seq, is_end, bonded, flat, box=None, is_rna=None):
        lib = load()
        seq = np.ascontiguousarray(seq, dtype=np.int32)
        self.n = int(seq.shape[0])
        is_end = np.zeros(self.n, np.uint8) if is_end is None else np.ascontiguousarray(is_end, dtype=np.uint8)
        bonded = np.ascontiguousarray(bonded, dtype=np.int32).reshape(-1, 2)
        flat = np.ascontiguousarray(flat, dtype=np.float64)
        box_arr = None if box is None else np.ascontiguousarray(np.broadcast_to(np.asarray(box, np.float64), (3,)))
        self._h = lib.mythos_cpu_create(int(model), self.n, seq.ctypes.data_as(_ip), is_end.ctypes.data_as(C.POINTER(C.c_uint8)),
                                        int(bonded.shape[0]), bonded.ctypes.data_as(_ip), None if box_arr is None else _d(box_arr),
                                        _d(flat), int(flat.shape[0]))
        if not self._h:
            raise ValueError("mythos_cpu_create: invalid arguments (parameter vector length, model)")
        self._lib = lib
        if is_rna is not None:  # oxNA (model 4; flat = the oxDNA2, oxRNA2 and hybrid vectors one after the other)
            t = np.ascontiguousarray(is_rna, dtype=np.uint8)
            lib.mythos_cpu_set_types(self._h, t.ctypes.data_as(C.POINTER(C.c_uint8)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.mythos_cpu_destroy(self._h)
            self._h = None

    def set_pairs(self, pairs) -> None:
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        self._lib.mythos_cpu_set_pairs(self._h, pairs.ctypes.data_as(_ip), int(pairs.shape[0]))

    def build_pairs(self, center, r_list: float) -> float:
        c = np.ascontiguousarray(center, dtype=np.float64)
        self._lib.mythos_cpu_build_pairs(self._h, _d(c), float(r_list))
        return float(self._lib.mythos_cpu_mean_row(self._h))

    def energy(self, center, quat):
        """-> (e_terms (8,), dU/dcenter (n,3), dU/dquat (n,4), body torque (n,3))"""
        c = np.ascontiguousarray(center, dtype=np.float64)
        q = np.ascontiguousarray(quat, dtype=np.float64)
        e, gc, gq, tb = np.zeros(8), np.zeros((self.n, 3)), np.zeros((self.n, 4)), np.zeros((self.n, 3))
        self._lib.mythos_cpu_energy(self._h, _d(c), _d(q), _d(e), _d(gc), _d(gq), _d(tb))
        return e, gc, gq, tb

    def run(self, c, q, p, L, n_steps, *, dt, kT, gamma_t, gamma_r, mass=1.0, inertia=(1.0, 1.0, 1.0), seed=0, step0=0,
            r_cut=0.0, skin=0.0, rebuild_every=0):
        """In place on contiguous float64 arrays; returns (list builds, e_last (10,))."""
        for a in (c, q, p, L):
            assert a.dtype == np.float64 and a.flags.c_contiguous
        inertia = np.ascontiguousarray(inertia, dtype=np.float64)
        e = np.zeros(10)
        builds = self._lib.mythos_cpu_langevin_run(self._h, _d(c), _d(q), _d(p), _d(L), int(n_steps), float(dt), float(kT), float(gamma_t),
                                                   float(gamma_r), float(mass), _d(inertia), C.c_uint64(int(seed) & (2**64 - 1)),
                                                   int(step0), float(r_cut), float(skin), int(rebuild_every), _d(e))
        return int(builds), e


def threads() -> int:
    return int(load().mythos_cpu_threads())


def set_threads(n: int) -> None:
    load().mythos_cpu_set_threads(int(n))
