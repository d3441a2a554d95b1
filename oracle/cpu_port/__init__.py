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
        lib.mythos_cpu_martini_create.restype = V
        lib.mythos_cpu_martini_create.argtypes = [C.c_int, _ip, C.c_int, _dp, _dp, C.c_int, _ip, _dp, _dp, C.c_int, _ip, _dp, _dp, C.c_int,
                                                  C.c_double, _dp]
        lib.mythos_cpu_martini_destroy.argtypes = [V]
        lib.mythos_cpu_martini_destroy.restype = None
        lib.mythos_cpu_martini_energy.argtypes = [V, _dp, _dp, _dp, _dp]
        lib.mythos_cpu_martini_energy.restype = None
        lib.mythos_cpu_martini_run.argtypes = [V, _dp, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_int64,
                                               C.c_double, C.c_int, _dp]
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


class MartiniCpuPort:
    """One MARTINI system on the host (oracle/cpu_port/martini_cpu.cpp): LJ over a Verlet list, bonds, angles, BAOAB
    Langevin; fp64, OpenMP.  ``angle_kind`` 0 = G96 cosine (MARTINI 2), 1 = harmonic (MARTINI 3)."""

    def __init__(self, types, sigma, eps, bonds, bond_k, bond_r0, angles, angle_k, angle_t0, angle_kind=0, r_cut=1.1, mass=None):
        lib = load()
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        types, bonds, angles = i32(types), i32(bonds).reshape(-1, 2), i32(angles).reshape(-1, 3)
        sigma, eps = f64(sigma), f64(eps)
        self.n = int(types.shape[0])
        mass = None if mass is None else f64(mass)
        ip = lambda a: a.ctypes.data_as(_ip)  # noqa: E731
        self._h = lib.mythos_cpu_martini_create(self.n, ip(types), int(sigma.shape[0]), _d(sigma), _d(eps), int(bonds.shape[0]), ip(bonds),
                                                _d(f64(bond_k)), _d(f64(bond_r0)), int(angles.shape[0]), ip(angles), _d(f64(angle_k)),
                                                _d(f64(angle_t0)), int(angle_kind), float(r_cut), None if mass is None else _d(mass))
        if not self._h:
            raise ValueError("mythos_cpu_martini_create: invalid arguments")
        self._lib = lib

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.mythos_cpu_martini_destroy(self._h)
            self._h = None

    def energy(self, x, box):
        """-> ([lj, bond, angle], dU/dx (n, 3))"""
        x, box = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(box, dtype=np.float64).reshape(3)
        e, g = np.zeros(3), np.zeros((self.n, 3))
        self._lib.mythos_cpu_martini_energy(self._h, _d(x), _d(box), _d(e), _d(g))
        return e, g

    def run(self, x, v, box, n_steps, *, dt, kT, gamma, seed=0, step0=0, skin=0.3, rebuild_every=5):
        """In place on contiguous float64 arrays; returns (list builds, [lj, bond, angle, kinetic] of the final state)."""
        for a in (x, v):
            assert a.dtype == np.float64 and a.flags.c_contiguous
        box = np.ascontiguousarray(box, dtype=np.float64).reshape(3)
        e = np.zeros(4)
        builds = self._lib.mythos_cpu_martini_run(self._h, _d(x), _d(v), _d(box), int(n_steps), float(dt), float(kT), float(gamma),
                                                  C.c_uint64(int(seed) & (2**64 - 1)), int(step0), float(skin), int(rebuild_every), _d(e))
        return int(builds), e
