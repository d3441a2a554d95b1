"""Thin object wrappers over the C ABI handles (device memory is owned by torch tensors).

``OxdnaSystem``  <-> mythos_system_t   (topology + parameters + neighbour rows)
``LangevinIntegrator`` <-> mythos_sim_t

These are plumbing: they validate shapes/dtypes/devices and forward pointers.  The reference-
shaped API (EnergyFunction / Simulator protocols) lives in ``mythos_amd.energy`` and
``mythos_amd.simulators`` on top of them.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from mythos_amd import _lib

N_TERMS = 8
TRACE_WIDTH = 10
TERM_NAMES = (
    "fene",
    "bonded_excluded_volume",
    "stacking",
    "unbonded_excluded_volume",
    "hydrogen_bonding",
    "cross_stacking",
    "coaxial_stacking",
    "debye",
)


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return 0
    if dtype == torch.float64:
        return 1
    raise ValueError(f"unsupported dtype {dtype}: use torch.float32 or torch.float64")


class OxdnaSystem:
    """One oxDNA system on one GPU."""

    def __init__(self, model: int, seq, is_end, bonded, box=None, dtype=torch.float32, device=None, is_rna=None):
        lib = _lib.load()
        if _lib.device_count() == 0 or not torch.cuda.is_available():
            raise _lib.MythosHipError("no HIP device visible: the mythos_amd HIP path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("OxdnaSystem needs a cuda (HIP) device")
        self.dtype = dtype
        self.model = int(model)
        seq = np.ascontiguousarray(seq, dtype=np.int32)
        self.n = int(seq.shape[0])
        is_end = np.zeros(self.n, np.uint8) if is_end is None else np.ascontiguousarray(is_end, dtype=np.uint8)
        bonded = np.ascontiguousarray(bonded, dtype=np.int32).reshape(-1, 2)
        box_arr = None if box is None else np.ascontiguousarray(np.broadcast_to(np.asarray(box, np.float64), (3,)))
        self.box = box_arr
        self._h = lib.mythos_oxdna_create(
            self.model,
            self.n,
            seq.ctypes.data_as(_lib.c_int_p),
            is_end.ctypes.data_as(_lib.c_uint8_p),
            int(bonded.shape[0]),
            bonded.ctypes.data_as(_lib.c_int_p),
            None if box_arr is None else box_arr.ctypes.data_as(_lib.c_double_p),
            _dtype_code(dtype),
            self.device.index or 0,
        )
        if not self._h:
            raise _lib.MythosHipError(f"mythos_oxdna_create: {_lib.last_error()}")
        self._lib = lib
        self._pseq_n_bp = 0
        self._pseq_terms = 0
        # oxNA (model 4): three vectors - oxDNA2, oxRNA2, hybrid - one after the other; dU/dparams rows likewise
        self.n_params = lib.mythos_oxdna_param_count() * (3 if self.model == 4 else 1)
        if self.model == 4:
            if is_rna is None:
                raise ValueError("an oxNA system (model 4) needs is_rna, the type of every nucleotide")
            t = np.ascontiguousarray(is_rna, dtype=np.uint8)
            if t.shape != (self.n,):
                raise ValueError(f"is_rna must have shape ({self.n},)")
            _lib.check(lib.mythos_oxdna_set_nucleotide_types(self._h, t.ctypes.data_as(_lib.c_uint8_p)), "set_nucleotide_types")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mythos_oxdna_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    # ---- parameters / neighbours -------------------------------------------------------------
    def set_params(self, flat) -> None:
        flat = np.ascontiguousarray(torch.as_tensor(flat).detach().cpu().numpy(), dtype=np.float64)
        _lib.check(
            self._lib.mythos_oxdna_set_params(self._h, flat.ctypes.data_as(_lib.c_double_p), int(flat.shape[0])),
            "set_params",
        )

    def set_pseq(self, marginals=None, unit=None, bp_probs=None, terms: int = 0) -> None:
        """Probabilistic sequence (mythos_oxdna_set_pseq); ``terms`` = 0 or no arguments: back to the discrete sequence."""
        if not terms:
            _lib.check(self._lib.mythos_oxdna_set_pseq(self._h, None, None, 0, None, 0), "set_pseq")
            self._pseq_terms = 0
            return
        marg = np.ascontiguousarray(marginals, dtype=np.float64)
        unit = np.ascontiguousarray(unit, dtype=np.int32)
        bp = np.ascontiguousarray(bp_probs, dtype=np.float64).reshape(-1, 4)
        if marg.shape != (self.n, 4) or unit.shape != (self.n,):
            raise ValueError(f"marginals must be ({self.n}, 4) and unit ({self.n},)")
        # every row of bp_probs is a constrained base pair (the library checks that unit names each of them exactly
        # twice, once per member); a system without base pairs passes one row of zeros, as the reference does
        n_bp = int(bp.shape[0]) if (unit >= 0).any() else 0
        self._pseq_n_bp = n_bp
        if (unit >= 0).any() and int(unit.max()) >= 2 * n_bp:
            raise ValueError("bp_probs has fewer rows than the base pairs named in unit")
        _lib.check(self._lib.mythos_oxdna_set_pseq(self._h, marg.ctypes.data_as(_lib.c_double_p), unit.ctypes.data_as(_lib.c_int_p),
                                                   n_bp, bp.ctypes.data_as(_lib.c_double_p), int(terms)), "set_pseq")
        self._pseq_terms = int(terms)

    def set_neighbors(self, pairs) -> None:
        pairs = np.ascontiguousarray(pairs, dtype=np.int32)
        if pairs.ndim != 2 or (pairs.size and pairs.shape[1] != 2):
            raise ValueError("pairs must have shape (P, 2)")
        _lib.check(
            self._lib.mythos_oxdna_set_neighbors(self._h, pairs.ctypes.data_as(_lib.c_int_p), int(pairs.shape[0])),
            "set_neighbors",
        )

    def build_neighbors(self, center: torch.Tensor, r_cut: float, skin: float) -> None:
        c = self._check(center, (self.n, 3), "center")
        _lib.check(
            self._lib.mythos_oxdna_build_neighbors(self._h, _lib.ptr(c), float(r_cut), float(skin), _stream(self.device)),
            "build_neighbors",
        )

    def neighbor_stats(self) -> tuple[int, float]:
        mx, mean = C.c_int(0), C.c_double(0.0)
        _lib.check(self._lib.mythos_oxdna_neighbor_stats(self._h, C.byref(mx), C.byref(mean)), "neighbor_stats")
        return mx.value, mean.value

    # ---- energy --------------------------------------------------------------------------------
    def _check(self, t: torch.Tensor, tail: tuple, name: str) -> torch.Tensor:
        if not isinstance(t, torch.Tensor) or t.device != self.device:
            raise ValueError(f"{name} must be a torch tensor on {self.device}")
        if t.dtype != self.dtype:
            raise ValueError(f"{name} must have dtype {self.dtype}, got {t.dtype}")
        if tuple(t.shape[-len(tail):]) != tail:
            raise ValueError(f"{name} must have trailing shape {tail}, got {tuple(t.shape)}")
        return t.contiguous()

    def energy(self, center, quat, *, grads=False, param_grads=False, observables=None, pseq_grads=False):
        """Term energies (F, 8) [float64] and optionally dU/dcenter, dU/dquat, dU/dflat.

        ``center`` (F, N, 3) or (N, 3); ``quat`` likewise with 4.  ``observables``: an
        ``mythos_amd.observables.ObservableSet`` evaluated in the same call (the observables kernel queued right behind the energy launch); its (F, width) rows are
        then returned as a fifth value.  ``pseq_grads`` (with ``param_grads``, after ``set_pseq``): also
        dU/d(marginals) (F, N, 4) and dU/d(base-pair type probabilities) (F, max(n_bp, 1), 4), as a fifth and sixth value.
        """
        single = center.dim() == 2
        c = self._check(center, (self.n, 3), "center")
        q = self._check(quat, (self.n, 4), "quat")
        if single:
            c, q = c[None], q[None]
        nf = c.shape[0]
        if q.shape[0] != nf:
            raise ValueError("center and quat disagree on the number of frames")
        e = torch.empty((nf, N_TERMS), dtype=torch.float64, device=self.device)
        gc = torch.empty_like(c) if grads else None
        gq = torch.empty_like(q) if grads else None
        gp = torch.empty((nf, self.n_params), dtype=torch.float64, device=self.device) if param_grads else None
        if pseq_grads:
            if not param_grads or observables is not None:
                raise ValueError("pseq_grads comes with param_grads and without fused observables")
            gm = torch.empty((nf, self.n, 4), dtype=torch.float64, device=self.device)
            gb = torch.empty((nf, max(self._pseq_n_bp, 1), 4), dtype=torch.float64, device=self.device)
            _lib.check(
                self._lib.mythos_oxdna_energy_dpseq(
                    self._h, _lib.ptr(c), _lib.ptr(q), nf, _lib.ptr(e), _lib.ptr(gc), _lib.ptr(gq), _lib.ptr(gp), _lib.ptr(gm),
                    _lib.ptr(gb), _stream(self.device),
                ),
                "energy_dpseq",
            )
            if single:
                return e[0], (gc[0] if grads else None), (gq[0] if grads else None), gp[0], gm[0], gb[0]
            return e, gc, gq, gp, gm, gb
        if observables is None:
            _lib.check(
                self._lib.mythos_oxdna_energy(
                    self._h, _lib.ptr(c), _lib.ptr(q), nf, _lib.ptr(e), _lib.ptr(gc), _lib.ptr(gq), _lib.ptr(gp),
                    _stream(self.device),
                ),
                "energy",
            )
            rows = None
        else:
            rows = torch.empty((nf, observables.width), dtype=torch.float64, device=self.device)
            _lib.check(
                self._lib.mythos_oxdna_energy_obs(
                    self._h, _lib.ptr(c), _lib.ptr(q), nf, _lib.ptr(e), _lib.ptr(gc), _lib.ptr(gq), _lib.ptr(gp),
                    observables._h, _lib.ptr(rows), _stream(self.device),
                ),
                "energy_obs",
            )
        if single:
            e = e[0]
            gc = gc[0] if grads else None
            gq = gq[0] if grads else None
            gp = gp[0] if param_grads else None
        if observables is not None:
            return e, gc, gq, gp, rows
        return e, gc, gq, gp


def _touched(*tensors) -> None:
    """The library has just written into these caller tensors through raw pointers: bump torch's version counters, so that
    anything keyed on them (the fused-observable rows of mythos_amd/observables/base.py) sees the change."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


class LangevinIntegrator:
    """BAOAB rigid-body Langevin dynamics bound to an :class:`OxdnaSystem`."""

    def __init__(self, system: OxdnaSystem, dt, kT, gamma_t, gamma_r, mass=1.0, inertia=(1.0, 1.0, 1.0), seed=0):
        self.system = system
        self._lib = system._lib
        inertia = np.ascontiguousarray(inertia, dtype=np.float64)
        self._h = self._lib.mythos_langevin_create(
            system._h, float(dt), float(kT), float(gamma_t), float(gamma_r), float(mass),
            inertia.ctypes.data_as(_lib.c_double_p), C.c_uint64(int(seed) & (2**64 - 1)),
        )
        if not self._h:
            raise _lib.MythosHipError(f"mythos_langevin_create: {_lib.last_error()}")
        self.dt, self.kT = float(dt), float(kT)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mythos_langevin_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def set_neighbor_policy(self, r_cut: float, skin: float, every: int) -> None:
        _lib.check(
            self._lib.mythos_langevin_set_neighbor_policy(self._h, float(r_cut), float(skin), int(every)),
            "set_neighbor_policy",
        )

    def set_unfused(self, on: bool = True) -> None:
        """oxNA systems: step through the two-launch path (forces launch + integrator launch) from the next load / run
        on - the second implementation the fused oxNA step kernel is checked against (mythos_langevin_set_option)."""
        _lib.check(self._lib.mythos_langevin_set_option(self._h, 0, 1 if on else 0), "set_option(unfused)")

    def init_momenta(self):
        s = self.system
        p = torch.empty((s.n, 3), dtype=s.dtype, device=s.device)
        ang = torch.empty((s.n, 3), dtype=s.dtype, device=s.device)
        _lib.check(self._lib.mythos_langevin_init_momenta(self._h, _lib.ptr(p), _lib.ptr(ang), _stream(s.device)), "init_momenta")
        return p, ang

    def run(self, center, quat, p_lin, p_ang, n_steps: int, save_every: int = 0, want_energy: bool = True):
        """Advance in place; returns (traj_center, traj_quat, e_trace) or Nones when save_every == 0."""
        s = self.system
        for t, tail, name in ((center, (s.n, 3), "center"), (quat, (s.n, 4), "quat"), (p_lin, (s.n, 3), "p_lin"), (p_ang, (s.n, 3), "p_ang")):
            if s._check(t, tail, name).data_ptr() != t.data_ptr():
                raise ValueError(f"{name} must be contiguous (it is updated in place)")
        n_save = n_steps // save_every if save_every > 0 else 0
        tc = torch.empty((n_save, s.n, 3), dtype=s.dtype, device=s.device) if n_save else None
        tq = torch.empty((n_save, s.n, 4), dtype=s.dtype, device=s.device) if n_save else None
        et = torch.zeros((n_save, TRACE_WIDTH), dtype=torch.float64, device=s.device) if (n_save and want_energy) else None
        rc = self._lib.mythos_langevin_run(
            self._h, _lib.ptr(center), _lib.ptr(quat), _lib.ptr(p_lin), _lib.ptr(p_ang), int(n_steps),
            int(save_every), _lib.ptr(tc), _lib.ptr(tq), _lib.ptr(et), _stream(s.device),
        )
        _touched(center, quat, p_lin, p_ang)  # (a run that fails still hands back the state of its last valid step)
        _lib.check(rc, "langevin_run")
        return tc, tq, et

    # ---- resident form: the state stays in the integrator's layout on the device between calls ----------
    def _state_ptrs(self, center, quat, p_lin, p_ang):
        s = self.system
        for t, tail, name in ((center, (s.n, 3), "center"), (quat, (s.n, 4), "quat"), (p_lin, (s.n, 3), "p_lin"), (p_ang, (s.n, 3), "p_ang")):
            if s._check(t, tail, name).data_ptr() != t.data_ptr():
                raise ValueError(f"{name} must be contiguous")
        return _lib.ptr(center), _lib.ptr(quat), _lib.ptr(p_lin), _lib.ptr(p_ang)

    def load(self, center, quat, p_lin, p_ang) -> None:
        """Copy a state into the integrator (mythos_langevin_load); ``advance`` then steps it in place."""
        _lib.check(self._lib.mythos_langevin_load(self._h, *self._state_ptrs(center, quat, p_lin, p_ang), _stream(self.system.device)), "langevin_load")

    def advance(self, n_steps: int, save_every: int = 0, want_energy: bool = True, out=None):
        """``n_steps`` on the resident state; the neighbour list and its rebuild schedule carry over between calls.
        Returns (traj_center, traj_quat, e_trace) or Nones when save_every == 0.  ``out = (traj_center, traj_quat)``:
        rows written into the caller's tensors ((n_steps // save_every, n, 3 | 4), contiguous) instead of new ones."""
        s = self.system
        n_save = n_steps // save_every if save_every > 0 else 0
        if out is not None and n_save:
            tc, tq = out
            for t, w in ((tc, 3), (tq, 4)):
                if t.device != s.device or t.dtype != s.dtype or tuple(t.shape) != (n_save, s.n, w) or not t.is_contiguous():
                    raise ValueError(f"out tensors must be contiguous {s.dtype} of shape ({n_save}, {s.n}, 3) and ({n_save}, {s.n}, 4) on {s.device}")
        else:
            tc = torch.empty((n_save, s.n, 3), dtype=s.dtype, device=s.device) if n_save else None
            tq = torch.empty((n_save, s.n, 4), dtype=s.dtype, device=s.device) if n_save else None
        et = torch.zeros((n_save, TRACE_WIDTH), dtype=torch.float64, device=s.device) if (n_save and want_energy) else None
        _lib.check(
            self._lib.mythos_langevin_advance(self._h, int(n_steps), int(save_every), _lib.ptr(tc), _lib.ptr(tq), _lib.ptr(et), _stream(s.device)),
            "langevin_advance",
        )
        if out is not None and n_save:
            _touched(tc, tq)
        return tc, tq, et

    def store(self, center, quat, p_lin, p_ang) -> None:
        """Copy the resident state out (mythos_langevin_store, asynchronous on the current stream)."""
        rc = self._lib.mythos_langevin_store(self._h, *self._state_ptrs(center, quat, p_lin, p_ang), _stream(self.system.device))
        _touched(center, quat, p_lin, p_ang)  # (the arrays are written even when closing an open frame failed)
        _lib.check(rc, "langevin_store")

    @property
    def step(self) -> int:
        return int(self._lib.mythos_langevin_get_step(self._h))

    @step.setter
    def step(self, value: int) -> None:
        _lib.check(self._lib.mythos_langevin_set_step(self._h, int(value)), "set_step")

    def set_seed(self, seed: int) -> None:
        """Key of the noise from the next launch / init_momenta on (mythos_langevin_set_seed)."""
        _lib.check(self._lib.mythos_langevin_set_seed(self._h, C.c_uint64(int(seed) & (2**64 - 1))), "set_seed")

    def set_timing(self, samples: int) -> None:
        """Bracket ``samples`` dispatches per run with HIP event pairs (0 = off, the default; see last_kernel_ms)."""
        _lib.check(self._lib.mythos_langevin_set_timing(self._h, int(samples)), "set_timing")

    def last_recoveries(self) -> int:
        """Out-of-turn list rebuilds of the last run (a site left its skin early, or rows / buckets had to grow)."""
        r = C.c_int(0)
        _lib.check(self._lib.mythos_langevin_last_recoveries(self._h, C.byref(r)), "last_recoveries")
        return int(r.value)

    def last_rebuilds(self) -> int:
        """Scheduled list rebuilds inside the last run / advance."""
        r = C.c_int(0)
        _lib.check(self._lib.mythos_langevin_last_rebuilds(self._h, C.byref(r)), "last_rebuilds")
        return int(r.value)

    def last_kernel_ms(self) -> dict:
        """HIP-event timings of the last run (see include/mythos_hip.h)."""
        k, loop, n, ns = C.c_double(0.0), C.c_double(0.0), C.c_int(0), C.c_int(0)
        _lib.check(
            self._lib.mythos_langevin_last_kernel_ms(self._h, C.byref(k), C.byref(loop), C.byref(n), C.byref(ns)),
            "last_kernel_ms",
        )
        return {"kernel_ms": k.value, "loop_ms_per_launch": loop.value, "launches": n.value, "samples": ns.value}


class MartiniSystem:
    """One MARTINI system on one GPU (mythos_martini_t): LJ type tables, bonds, angles."""

    N_TERMS = 3  # lj, bond, angle

    def __init__(self, types, sigma, eps, bonds, bond_k, bond_r0, angles, angle_k, angle_t0, angle_kind=0, r_cut=1.1,
                 dtype=torch.float32, device=None):
        lib = _lib.load()
        if _lib.device_count() == 0 or not torch.cuda.is_available():
            raise _lib.MythosHipError("no HIP device visible: the mythos_amd HIP path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dtype = dtype
        types = np.ascontiguousarray(types, dtype=np.int32)
        self.n = int(types.shape[0])
        sigma = np.ascontiguousarray(sigma, dtype=np.float64)
        eps = np.ascontiguousarray(eps, dtype=np.float64)
        n_types = int(sigma.shape[0])
        bonds = np.ascontiguousarray(bonds, dtype=np.int32).reshape(-1, 2)
        angles = np.ascontiguousarray(angles, dtype=np.int32).reshape(-1, 3)
        self.n_types, self.n_bonds, self.n_angles = n_types, int(bonds.shape[0]), int(angles.shape[0])
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        bond_k, bond_r0, angle_k, angle_t0 = f(bond_k), f(bond_r0), f(angle_k), f(angle_t0)
        dp = lambda a: a.ctypes.data_as(_lib.c_double_p)  # noqa: E731
        self._h = lib.mythos_martini_create(
            self.n, types.ctypes.data_as(_lib.c_int_p), n_types, dp(sigma), dp(eps), int(bonds.shape[0]),
            bonds.ctypes.data_as(_lib.c_int_p), dp(bond_k), dp(bond_r0), int(angles.shape[0]),
            angles.ctypes.data_as(_lib.c_int_p), dp(angle_k), dp(angle_t0), int(angle_kind), float(r_cut),
            _dtype_code(dtype), self.device.index or 0,
        )
        if not self._h:
            raise _lib.MythosHipError(f"mythos_martini_create: {_lib.last_error()}")
        self._lib = lib

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mythos_martini_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def energy(self, pos: torch.Tensor, box: torch.Tensor, grads: bool = False):
        """pos (F, N, 3) or (N, 3); box (F, 3) or (3,) -> (e_terms (F, 3) float64, dU/dpos or None)."""
        single = pos.dim() == 2
        if single:
            pos, box = pos[None], torch.as_tensor(box).reshape(1, 3)
        if pos.device != self.device or pos.dtype != self.dtype or tuple(pos.shape[1:]) != (self.n, 3):
            raise ValueError(f"pos must be a {self.dtype} tensor of shape (F, {self.n}, 3) on {self.device}")
        pos = pos.contiguous()
        box = torch.as_tensor(box, dtype=self.dtype, device=self.device).reshape(-1, 3)
        if box.shape[0] == 1 and pos.shape[0] > 1:
            box = box.expand(pos.shape[0], 3)
        box = box.contiguous()
        nf = pos.shape[0]
        e = torch.empty((nf, 3), dtype=torch.float64, device=self.device)
        g = torch.empty_like(pos) if grads else None
        _lib.check(
            self._lib.mythos_martini_energy(self._h, _lib.ptr(pos), _lib.ptr(box), nf, _lib.ptr(e), _lib.ptr(g), _stream(self.device)),
            "martini_energy",
        )
        return (e[0], g[0] if grads else None) if single else (e, g)

    def param_grads(self, pos: torch.Tensor, box: torch.Tensor, lj: bool = True, bonds: bool = True, angles: bool = True):
        """Per-frame parameter gradients (float64, on the device): dict with ``sigma``/``eps`` (F, T, T) for the
        ordered type pair, ``bond_k``/``bond_r0`` (F, n_bonds), ``angle_k``/``angle_t0`` (F, n_angles)."""
        if pos.dim() == 2:
            pos, box = pos[None], torch.as_tensor(box).reshape(1, 3)
        if pos.device != self.device or pos.dtype != self.dtype or tuple(pos.shape[1:]) != (self.n, 3):
            raise ValueError(f"pos must be a {self.dtype} tensor of shape (F, {self.n}, 3) on {self.device}")
        pos = pos.contiguous()
        box = torch.as_tensor(box, dtype=self.dtype, device=self.device).reshape(-1, 3)
        if box.shape[0] == 1 and pos.shape[0] > 1:
            box = box.expand(pos.shape[0], 3)
        box = box.contiguous()
        nf = pos.shape[0]

        def buf(on, *shape):
            return torch.zeros((nf, *shape), dtype=torch.float64, device=self.device) if on else None

        out = {"sigma": buf(lj, self.n_types, self.n_types), "eps": buf(lj, self.n_types, self.n_types),
               "bond_k": buf(bonds, self.n_bonds), "bond_r0": buf(bonds, self.n_bonds),
               "angle_k": buf(angles, self.n_angles), "angle_t0": buf(angles, self.n_angles)}
        _lib.check(
            self._lib.mythos_martini_param_grads(
                self._h, _lib.ptr(pos), _lib.ptr(box), nf, _lib.ptr(out["sigma"]), _lib.ptr(out["eps"]),
                _lib.ptr(out["bond_k"]), _lib.ptr(out["bond_r0"]), _lib.ptr(out["angle_k"]), _lib.ptr(out["angle_t0"]),
                _stream(self.device)),
            "martini_param_grads",
        )
        return out


class MartiniLangevinIntegrator:
    """BAOAB Langevin dynamics of a :class:`MartiniSystem` (mythos_martini_sim_t): LJ over a device-built Verlet
    list, bonds, angles; units nm, ps, amu, kJ/mol.  ``gamma`` is the friction rate in 1/ps."""

    KB = 0.0083144626  # kJ/mol/K

    def __init__(self, system: MartiniSystem, dt, kT, gamma, mass=None, seed=0):
        self.system = system
        self._lib = system._lib
        mptr = None
        if mass is not None:
            mass = np.ascontiguousarray(mass, dtype=np.float64)
            if mass.shape != (system.n,):
                raise ValueError(f"mass must have shape ({system.n},)")
            mptr = mass.ctypes.data_as(_lib.c_double_p)
        self._h = self._lib.mythos_martini_langevin_create(system._h, float(dt), float(kT), float(gamma), mptr, int(seed))
        if not self._h:
            raise _lib.MythosHipError(f"mythos_martini_langevin_create: {_lib.last_error()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mythos_martini_langevin_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def set_neighbor_policy(self, skin: float, every: int) -> None:
        _lib.check(self._lib.mythos_martini_langevin_set_neighbor_policy(self._h, float(skin), int(every)), "set_neighbor_policy")

    def set_inner_list(self, margin: float, every: int) -> None:
        """Pruned rows inside r_cut + ``margin``, rewritten every ``every`` steps by the step launch (off by default;
        margin <= 0 switches them off again) - include/mythos_hip.h."""
        _lib.check(self._lib.mythos_martini_langevin_set_inner_list(self._h, float(margin), int(every)), "set_inner_list")

    def init_velocities(self) -> torch.Tensor:
        v = torch.empty((self.system.n, 3), dtype=self.system.dtype, device=self.system.device)
        _lib.check(self._lib.mythos_martini_langevin_init_velocities(self._h, _lib.ptr(v), _stream(self.system.device)), "init_velocities")
        return v

    def _check_state(self, pos, vel):
        s = self.system
        for t, name in ((pos, "pos"), (vel, "vel")):
            if t.device != s.device or t.dtype != s.dtype or tuple(t.shape) != (s.n, 3) or not t.is_contiguous():
                raise ValueError(f"{name} must be a contiguous {s.dtype} tensor of shape ({s.n}, 3) on {s.device}")

    def _rows(self, n_steps: int, save_every: int, want_energy: bool):
        s = self.system
        ns = n_steps // save_every if save_every > 0 else 0
        traj = torch.empty((ns, s.n, 3), dtype=s.dtype, device=s.device) if ns else None
        et = torch.zeros((ns, 4), dtype=torch.float64, device=s.device) if (ns and want_energy) else None
        return traj, et

    def run(self, pos, vel, box, n_steps: int, save_every: int = 0, want_energy: bool = True):
        """Advance ``pos`` / ``vel`` (n, 3) in place -> (traj_pos (S, n, 3) or None, e_trace (S, 4) float64 or None);
        e_trace columns: lj, bond, angle, kinetic (kJ/mol) at the saved steps (``want_energy=False``: positions only)."""
        s = self.system
        self._check_state(pos, vel)
        box = np.ascontiguousarray(np.asarray(box, dtype=np.float64).reshape(3))
        traj, et = self._rows(n_steps, save_every, want_energy)
        rc = self._lib.mythos_martini_langevin_run(
            self._h, _lib.ptr(pos), _lib.ptr(vel), box.ctypes.data_as(_lib.c_double_p), int(n_steps), int(save_every),
            _lib.ptr(traj), _lib.ptr(et), _stream(s.device))
        _touched(pos, vel)
        _lib.check(rc, "martini_langevin_run")
        return traj, et

    # ---- resident form (mythos_martini_langevin_load / advance / store) -----------------------------------
    def load(self, pos, vel, box) -> None:
        self._check_state(pos, vel)
        box = np.ascontiguousarray(np.asarray(box, dtype=np.float64).reshape(3))
        _lib.check(self._lib.mythos_martini_langevin_load(self._h, _lib.ptr(pos), _lib.ptr(vel), box.ctypes.data_as(_lib.c_double_p),
                                                          _stream(self.system.device)), "martini_langevin_load")

    def advance(self, n_steps: int, save_every: int = 0, want_energy: bool = True):
        """``n_steps`` on the resident state: n launches, the frame stays open; the list and its schedule carry over."""
        traj, et = self._rows(n_steps, save_every, want_energy)
        _lib.check(self._lib.mythos_martini_langevin_advance(self._h, int(n_steps), int(save_every), _lib.ptr(traj), _lib.ptr(et),
                                                             _stream(self.system.device)), "martini_langevin_advance")
        return traj, et

    def store(self, pos, vel) -> None:
        self._check_state(pos, vel)
        rc = self._lib.mythos_martini_langevin_store(self._h, _lib.ptr(pos), _lib.ptr(vel), _stream(self.system.device))
        _touched(pos, vel)
        _lib.check(rc, "martini_langevin_store")

    @property
    def step(self) -> int:
        return int(self._lib.mythos_martini_langevin_get_step(self._h))

    def last_rebuilds(self) -> int:
        r = C.c_int(0)
        _lib.check(self._lib.mythos_martini_langevin_last_rebuilds(self._h, C.byref(r)), "last_rebuilds")
        return int(r.value)

    def set_timing(self, samples: int) -> None:
        """Bracket ``samples`` dispatches per run with HIP event pairs (0 = off, the default; see last_kernel_ms)."""
        _lib.check(self._lib.mythos_martini_langevin_set_timing(self._h, int(samples)), "set_timing")

    def last_recoveries(self) -> int:
        """Out-of-turn list rebuilds of the last run (a bead left its skin early, or rows / buckets had to grow)."""
        r = C.c_int(0)
        _lib.check(self._lib.mythos_martini_langevin_last_recoveries(self._h, C.byref(r)), "last_recoveries")
        return int(r.value)

    def last_kernel_ms(self) -> dict:
        k, loop, n, ns = C.c_double(0.0), C.c_double(0.0), C.c_int(0), C.c_int(0)
        _lib.check(self._lib.mythos_martini_langevin_last_kernel_ms(self._h, C.byref(k), C.byref(loop), C.byref(n), C.byref(ns)),
                   "last_kernel_ms")
        return {"kernel_ms": k.value, "loop_ms_per_launch": loop.value, "launches": n.value, "samples": ns.value}

    def rows(self, pruned: bool = False):
        """(rows (n, stride) int32, lengths (n,) int32) of the Verlet rows, or of the pruned rows, as numpy arrays
        (diagnostics / tests; synchronises)."""
        stride = C.c_int(0)
        which = 1 if pruned else 0
        _lib.check(self._lib.mythos_martini_langevin_get_rows(self._h, which, None, None, C.byref(stride)), "get_rows")
        rows = np.empty((self.system.n, stride.value), dtype=np.int32)
        lens = np.empty(self.system.n, dtype=np.int32)
        _lib.check(self._lib.mythos_martini_langevin_get_rows(self._h, which, rows.ctypes.data_as(C.POINTER(C.c_int32)),
                                                              lens.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(stride)), "get_rows")
        return rows, lens

    def neighbor_stats(self) -> tuple[int, float]:
        mx, mean = C.c_int(0), C.c_double(0.0)
        _lib.check(self._lib.mythos_martini_langevin_neighbor_stats(self._h, C.byref(mx), C.byref(mean)), "neighbor_stats")
        return mx.value, mean.value
