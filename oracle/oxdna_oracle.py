"""CPU oracle for the oxDNA1 / oxDNA2 energy path  --  TEST INFRASTRUCTURE ONLY.

This file is a torch-fp64 restatement of the reference's JAX arithmetic.  It is the
checker for the HIP kernels (energies, forces, torques, dU/dq, dU/dtheta through
``torch.autograd``) and the timed ``cpu_baseline`` of ``bench.py``.  Nothing in the
product package (``mythos_amd/``) imports it; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may.

Parity pin: the restatement is checked in ``tests/test_oracle_golden.py`` against the
oxDNA-standalone golden files the reference's own tests use
(``tests/golden/dna{1,2}/*/split_energy.dat``, ``energy.dat``, and the per-pair
``pair.dat``) with the reference's own tolerances.  The reference itself cannot be
imported here (jax / jax_md / chex absent: ordinary ModuleNotFoundError), so no vectors
were generated from it.  Forces, torques, dU/dtheta and the integrator are NOT pinned by
any reference test; for those this oracle's autograd and central
finite differences are the stand-in.

Every function cites the reference file:line it follows (paths relative to the reference
root, ``mythos/...``).
"""

from __future__ import annotations

import math

import torch

F64 = torch.float64
PI = math.pi

TERMS_DNA1 = (
    "fene",
    "bonded_excluded_volume",
    "stacking",
    "unbonded_excluded_volume",
    "hydrogen_bonding",
    "cross_stacking",
    "coaxial_stacking",
)
TERMS_DNA2 = (*TERMS_DNA1, "debye")


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=F64)


# ---------------------------------------------------------------------------------------------
# primitive forms: mythos/energy/potentials.py:11-70, mythos/utils/math.py:68-81
# ---------------------------------------------------------------------------------------------


def clamp(x, lo=-1.0, hi=1.0):
    """mythos/utils/math.py:78-81 (gradient is zero where clipped)."""
    hi_t = torch.full_like(x, hi)
    lo_t = torch.full_like(x, lo)
    c = torch.where(x >= hi, hi_t, x)
    return torch.where(c <= lo, lo_t, c)


def smooth_abs(x, eps=1e-10):
    """mythos/utils/math.py:68-75."""
    return torch.sqrt(x * x + eps)


def v_fene(r, eps, r0, delt):
    """potentials.py:11-19."""
    x = (r - r0) ** 2 / delt**2
    return -eps / 2.0 * torch.log(1 - x)


def v_morse(r, eps, r0, a):
    """potentials.py:22-33."""
    return eps * (1 - torch.exp(-(r - r0) * a)) ** 2


def v_harmonic(r, k, r0):
    """potentials.py:36-45."""
    return k / 2 * (r - r0) ** 2


def v_lj(r, eps, sigma):
    """potentials.py:48-54."""
    x = (sigma / r) ** 12 - (sigma / r) ** 6
    return 4 * eps * x


def v_mod(theta, a, theta0):
    """potentials.py:57-62."""
    return 1 - a * (theta - theta0) ** 2


def v_smooth(x, b, x_c):
    """potentials.py:65-70."""
    return b * (x_c - x) ** 2


# ---------------------------------------------------------------------------------------------
# f1..f6: mythos/energy/dna1/base_functions.py:13-129, mythos/energy/dna2/base_functions.py:13-17
# (all comparisons strict, exactly as the reference)
# ---------------------------------------------------------------------------------------------


def _where(cond, a, b):
    a = _t(a)
    b = _t(b)
    return torch.where(cond, a, b)


def f1(r, r_low, r_high, r_c_low, r_c_high, eps, a, r0, r_c, b_low, b_high):
    """base_functions.py:13-37."""
    zero = torch.zeros_like(r)
    oob = _where(
        (r_c_low < r) & (r < r_low),
        eps * v_smooth(r, b_low, r_c_low),
        _where((r_high < r) & (r < r_c_high), eps * v_smooth(r, b_high, r_c_high), zero),
    )
    core = v_morse(r, eps, r0, a) - v_morse(_t(r_c), eps, r0, a)
    return _where((r_low < r) & (r < r_high), core, oob)


def f2(r, r_low, r_high, r_c_low, r_c_high, k, r0, r_c, b_low, b_high):
    """base_functions.py:40-63."""
    zero = torch.zeros_like(r)
    oob = _where(
        (r_c_low < r) & (r < r_low),
        k * v_smooth(r, b_low, r_c_low),
        _where((r_high < r) & (r < r_c_high), k * v_smooth(r, b_high, r_c_high), zero),
    )
    core = v_harmonic(r, k, r0) - v_harmonic(_t(r_c), k, r0)
    return _where((r_low < r) & (r < r_high), core, oob)


def f3(r, r_star, r_c, eps, sigma, b):
    """base_functions.py:66-79.  The LJ branch is evaluated on a sanitised radius so the
    untaken branch never produces inf*0 in autograd (value identical where taken)."""
    zero = torch.zeros_like(r)
    oob = _where((r_star < r) & (r < r_c), eps * v_smooth(r, b, r_c), zero)
    inside = r < r_star
    r_safe = torch.where(inside, r, torch.ones_like(r))
    return _where(inside, v_lj(r_safe, eps, sigma), oob)


def f4(theta, theta0, delta_theta_star, delta_theta_c, a, b):
    """base_functions.py:82-107."""
    zero = torch.zeros_like(theta)
    oob = _where(
        (theta0 - delta_theta_c < theta) & (theta < theta0 - delta_theta_star),
        v_smooth(theta, b, theta0 - delta_theta_c),
        _where(
            (theta0 + delta_theta_star < theta) & (theta < theta0 + delta_theta_c),
            v_smooth(theta, b, theta0 + delta_theta_c),
            zero,
        ),
    )
    return _where(
        (theta0 - delta_theta_star < theta) & (theta < theta0 + delta_theta_star),
        v_mod(theta, a, theta0),
        oob,
    )


def f5(x, x_star, x_c, a, b):
    """base_functions.py:110-129."""
    one = torch.ones_like(x)
    zero = torch.zeros_like(x)
    return _where(
        x > 0.0,
        one,
        _where(
            (x_star < x) & (x < 0.0),
            v_mod(x, a, 0.0),
            _where((x_c < x) & (x < x_star), v_smooth(x, b, x_c), zero),
        ),
    )


def f6(theta, a, b):
    """dna2/base_functions.py:13-17."""
    return _where(theta >= b, a / 2 * (theta - b) ** 2, torch.zeros_like(theta))


# ---------------------------------------------------------------------------------------------
# smoothing-parameter solvers: mythos/energy/dna1/base_smoothing_functions.py:13-142
# ---------------------------------------------------------------------------------------------


def _solve_f1_b(x, a, x0, xc):
    """base_smoothing_functions.py:13-31."""
    e = torch.exp
    num = a**2 * (-e(a * (3 * x0 + 2 * xc)) + 2 * e(a * (x + 2 * x0 + 2 * xc)) - e(a * (2 * x + x0 + 2 * xc)))
    num = num * e(-2 * a * x)
    den = 2 * e(a * (x + 2 * xc)) + e(a * (2 * x + x0)) - 2 * e(a * (2 * x + xc)) - e(a * (x0 + 2 * xc))
    return num / den


def _solve_f1_xc_star(x, a, x0, xc):
    """base_smoothing_functions.py:34-47."""
    e = torch.exp
    num = (
        a * x * e(a * (x + 2 * xc))
        - a * x * e(a * (x0 + 2 * xc))
        + 2 * e(a * (x + 2 * xc))
        + e(a * (2 * x + x0))
        - 2 * e(a * (2 * x + xc))
        - e(a * (x0 + 2 * xc))
    ) * e(-2 * a * xc)
    return num / (a * (e(a * x) - e(a * x0)))


def get_f1_smoothing_params(x0, a, xc, x_low, x_high):
    """base_smoothing_functions.py:50-59 -> (b_low, xc_low, b_high, xc_high)."""
    x0, a, xc, x_low, x_high = map(_t, (x0, a, xc, x_low, x_high))
    return (
        _solve_f1_b(x_low, a, x0, xc),
        _solve_f1_xc_star(x_low, a, x0, xc),
        _solve_f1_b(x_high, a, x0, xc),
        _solve_f1_xc_star(x_high, a, x0, xc),
    )


def get_f2_smoothing_params(x0, xc, x_low, x_high):
    """base_smoothing_functions.py:62-81."""
    x0, xc, x_low, x_high = map(_t, (x0, xc, x_low, x_high))

    def b(x):
        return (x - x0) ** 2 / (2 * (x - xc) * (x - 2 * x0 + xc))

    def xcs(x):
        return (x * x0 - 2 * x0 * xc + xc**2) / (x - x0)

    return b(x_low), xcs(x_low), b(x_high), xcs(x_high)


def get_f3_smoothing_params(r_star, sigma):
    """base_smoothing_functions.py:84-104 -> (b, r_c)."""
    x, s = _t(r_star), _t(sigma)
    b = (
        -36
        * s**6
        * (-2 * s**6 + x**6) ** 2
        / (x**14 * (-s + x) * (s + x) * (s**2 - s * x + x**2) * (s**2 + s * x + x**2))
    )
    xc = x * (-7 * s**6 + 4 * x**6) / (3 * (-2 * s**6 + x**6))
    return b, xc


def get_f4_smoothing_params(a, x0, delta_x_star):
    """base_smoothing_functions.py:107-123 -> (b, delta_theta_c)."""
    a, x0, d = map(_t, (a, x0, delta_x_star))
    x = x0 + d
    b = -(a**2) * (x - x0) ** 2 / (a * x**2 - 2 * a * x * x0 + a * x0**2 - 1)
    xc = (-a * x * x0 + a * x0**2 - 1) / (a * (-x + x0))
    return b, xc - x0


def get_f5_smoothing_params(a, x_star):
    """base_smoothing_functions.py:126-142 (x0 = 0) -> (b, x_c)."""
    a, x = map(_t, (a, x_star))
    x0 = torch.zeros_like(a)
    b = -(a**2) * (x - x0) ** 2 / (a * x**2 - 2 * a * x * x0 + a * x0**2 - 1)
    xc = (a * x * x0 - a * x0**2 + 1) / (a * (x - x0))
    return b, xc


# ---------------------------------------------------------------------------------------------
# init_params per term (dependent constants)
# ---------------------------------------------------------------------------------------------

STACK_WEIGHTS_SA = torch.ones(4, 4, dtype=F64)  # dna1/stacking.py (uniform sequence-averaged weights)
HB_WEIGHTS_SA = torch.tensor(  # dna1/hydrogen_bonding.py:18-25
    [[0.0, 0.0, 0.0, 1.0], [0.0, 0.0, 1.0, 0.0], [0.0, 1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]], dtype=F64
)


def init_fene(p):
    """dna1/fene.py:26-28 (no dependent parameters)."""
    return {k: _t(v) for k, v in p.items()}


def init_exc_vol(p, with_backbone: bool):
    """dna1/bonded_excluded_volume.py:56-75, dna1/unbonded_excluded_volume.py:67-96."""
    q = {k: _t(v) for k, v in p.items()}
    q["b_base"], q["dr_c_base"] = get_f3_smoothing_params(q["dr_star_base"], q["sigma_base"])
    q["b_back_base"], q["dr_c_back_base"] = get_f3_smoothing_params(q["dr_star_back_base"], q["sigma_back_base"])
    q["b_base_back"], q["dr_c_base_back"] = get_f3_smoothing_params(q["dr_star_base_back"], q["sigma_base_back"])
    if with_backbone:
        q["b_backbone"], q["dr_c_backbone"] = get_f3_smoothing_params(q["dr_star_backbone"], q["sigma_backbone"])
    return q


def init_stacking(p):
    """dna1/stacking.py:120-183.  ``p`` must hold ``kt``; optional ``ss_stack_weights``."""
    q = {k: (v if k in ("pseq", "pseq_constraints") else (_t(v) if v is not None else None)) for k, v in p.items()}
    if q.get("ss_stack_weights") is None:
        q["eps_stack"] = (q["eps_stack_base"] + q["eps_stack_kt_coeff"] * q["kt"]) * STACK_WEIGHTS_SA
    else:
        q["eps_stack"] = q["ss_stack_weights"] * (
            1.0 - q["eps_stack_kt_coeff"] + (q["kt"] * 9.0 * q["eps_stack_kt_coeff"])
        )
    (q["b_low_stack"], q["dr_c_low_stack"], q["b_high_stack"], q["dr_c_high_stack"]) = get_f1_smoothing_params(
        q["dr0_stack"], q["a_stack"], q["dr_c_stack"], q["dr_low_stack"], q["dr_high_stack"]
    )
    # dna1 / dna2: theta 4, 5, 6; rna2 (rna2/stacking.py:60-176): theta 5, 6, 9, 10 - whichever the section holds
    for k in (4, 5, 6, 9, 10):
        if f"a_stack_{k}" not in q:
            continue
        q[f"b_stack_{k}"], q[f"delta_theta_stack_{k}_c"] = get_f4_smoothing_params(
            q[f"a_stack_{k}"], q[f"theta0_stack_{k}"], q[f"delta_theta_star_stack_{k}"]
        )
    for k in (1, 2):
        q[f"b_neg_cos_phi{k}_stack"], q[f"neg_cos_phi{k}_c_stack"] = get_f5_smoothing_params(
            q[f"a_stack_{k}"], q[f"neg_cos_phi{k}_star_stack"]
        )
    return q


def init_hydrogen_bonding(p):
    """dna1/hydrogen_bonding.py:148-223."""
    q = {k: (v if k in ("pseq", "pseq_constraints") else (_t(v) if v is not None else None)) for k, v in p.items()}
    if q.get("ss_hb_weights") is None:
        q["eps_hb_weights"] = HB_WEIGHTS_SA * q["eps_hb"]
    else:
        q["eps_hb_weights"] = q["ss_hb_weights"]
    (q["b_low_hb"], q["dr_c_low_hb"], q["b_high_hb"], q["dr_c_high_hb"]) = get_f1_smoothing_params(
        q["dr0_hb"], q["a_hb"], q["dr_c_hb"], q["dr_low_hb"], q["dr_high_hb"]
    )
    for k in (1, 2, 3, 4, 7, 8):
        q[f"b_hb_{k}"], q[f"delta_theta_hb_{k}_c"] = get_f4_smoothing_params(
            q[f"a_hb_{k}"], q[f"theta0_hb_{k}"], q[f"delta_theta_star_hb_{k}"]
        )
    return q


def init_cross_stacking(p):
    """dna1/cross_stacking.py:110-183."""
    q = {k: _t(v) for k, v in p.items()}
    (q["b_low_cross"], q["dr_c_low_cross"], q["b_high_cross"], q["dr_c_high_cross"]) = get_f2_smoothing_params(
        q["r0_cross"], q["dr_c_cross"], q["dr_low_cross"], q["dr_high_cross"]
    )
    for k in (1, 2, 3, 4, 7, 8):  # rna2/cross_stacking.py:97-147 has no theta4 block
        if f"a_cross_{k}" not in q:
            continue
        q[f"b_cross_{k}"], q[f"delta_theta_cross_{k}_c"] = get_f4_smoothing_params(
            q[f"a_cross_{k}"], q[f"theta0_cross_{k}"], q[f"delta_theta_star_cross_{k}"]
        )
    return q


def init_coaxial(p, model: int):
    """dna1/coaxial_stacking.py:106-172, dna2/coaxial_stacking.py:79-130."""
    q = {k: _t(v) for k, v in p.items()}
    (q["b_low_coax"], q["dr_c_low_coax"], q["b_high_coax"], q["dr_c_high_coax"]) = get_f2_smoothing_params(
        q["dr0_coax"], q["dr_c_coax"], q["dr_low_coax"], q["dr_high_coax"]
    )
    for k in (4, 1, 5, 6):
        q[f"b_coax_{k}"], q[f"delta_theta_coax_{k}_c"] = get_f4_smoothing_params(
            q[f"a_coax_{k}"], q[f"theta0_coax_{k}"], q[f"delta_theta_star_coax_{k}"]
        )
    if model == 1:
        q["b_cos_phi3_coax"], q["cos_phi3_c_coax"] = get_f5_smoothing_params(q["a_coax_3p"], q["cos_phi3_star_coax"])
        q["b_cos_phi4_coax"], q["cos_phi4_c_coax"] = get_f5_smoothing_params(q["a_coax_4p"], q["cos_phi4_star_coax"])
    return q


def init_debye(p):
    """dna2/debye.py:47-64.  ``p`` holds q_eff, lambda_factor, prefactor_coeff, kt, salt_conc,
    half_charged_ends."""
    q = {k: (_t(v) if k != "half_charged_ends" else bool(v)) for k, v in p.items()}
    lam = q["lambda_factor"] * torch.sqrt(q["kt"] / 0.1) / torch.sqrt(q["salt_conc"])
    kappa = 1.0 / lam
    r_high = 3 * lam
    pref = q["prefactor_coeff"] * (q["q_eff"] ** 2)
    smoothing = -(torch.exp(-r_high / lam) * pref * pref * (r_high + lam) * (r_high + lam)) / (
        -4.0 * r_high * r_high * r_high * lam * lam * pref
    )
    r_cut = r_high * (pref * r_high + 3.0 * pref * lam) / (pref * (r_high + lam))
    q.update(lambda_=lam, kappa=kappa, r_high=r_high, prefactor=pref, smoothing_coeff=smoothing, r_cut=r_cut)
    return q


def init_all(model: int, cfg: dict, kt, salt_conc=0.5, half_charged_ends=True) -> dict:
    """Dependent constants for every term of a model from TOML-shaped sections.

    ``cfg`` maps section name -> {param: value}; mirrors ``default_energy_configs``
    (dna1/__init__.py:27-55, dna2/__init__.py:33-71).
    """
    out = {
        "geometry": {k: _t(v) for k, v in cfg["geometry"].items()},
        "fene": init_fene(cfg["fene"]),
        "bonded_excluded_volume": init_exc_vol(cfg["bonded_excluded_volume"], with_backbone=False),
        "stacking": init_stacking({**cfg["stacking"], "kt": kt}),
        "unbonded_excluded_volume": init_exc_vol(cfg["unbonded_excluded_volume"], with_backbone=True),
        "hydrogen_bonding": init_hydrogen_bonding(cfg["hydrogen_bonding"]),
        "cross_stacking": init_cross_stacking(cfg["cross_stacking"]),
        # oxRNA2 composes the dna1 coaxial term (f5 of cos phi3 / phi4) with its own parameters
        # (rna2/tests/test_integration.py:258-287)
        "coaxial_stacking": init_coaxial(cfg["coaxial_stacking"], 1 if model == 3 else model),
    }
    if model in (2, 3):
        out["debye"] = init_debye(
            {**cfg["debye"], "kt": kt, "salt_conc": salt_conc, "half_charged_ends": half_charged_ends}
        )
    return out


# ---------------------------------------------------------------------------------------------
# geometry: mythos/energy/utils.py:18-36, dna1/nucleotide.py:29-53, dna2/nucleotide.py:30-58
# ---------------------------------------------------------------------------------------------


def quat_to_axes(q):
    """a1 (back-base), a2 (cross), a3 (normal) rows; un-normalised as in the reference."""
    q0, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    a1 = torch.stack([q0**2 + q1**2 - q2**2 - q3**2, 2 * (q1 * q2 + q0 * q3), 2 * (q1 * q3 - q0 * q2)], dim=-1)
    a2 = torch.stack([2 * (q1 * q2 - q0 * q3), q0**2 - q1**2 + q2**2 - q3**2, 2 * (q2 * q3 + q0 * q1)], dim=-1)
    a3 = torch.stack([2 * (q1 * q3 + q0 * q2), 2 * (q2 * q3 - q0 * q1), q0**2 - q1**2 - q2**2 + q3**2], dim=-1)
    return a1, a2, a3


class Sites:
    """Interaction sites of every nucleotide (one frame)."""

    def __init__(self, model: int, geometry: dict, center, a1, a2, a3):
        g = geometry
        self.center, self.a1, self.a2, self.a3 = center, a1, a2, a3
        if model != 3:
            self.stack = center + g["com_to_stacking"] * a1
            self.base = center + g["com_to_hb"] * a1
        if model == 3:
            # oxRNA2 (rna2/nucleotide.py:52-63): backbone offset on a1 and a3, stacking through separate 3' / 5' sites,
            # and two body-fixed vectors p3 / p5 towards the neighbouring phosphates
            self.stack = center + g["pos_stack"] * a1
            self.base = center + g["pos_base"] * a1
            self.back = center + g["pos_back_a1"] * a1 + g["pos_back_a3"] * a3
            self.back_dna1 = self.back
            self.p3 = g["p3_x"] * a1 + g["p3_y"] * a2 + g["p3_z"] * a3
            self.p5 = g["p5_x"] * a1 + g["p5_y"] * a2 + g["p5_z"] * a3
            self.stack3 = center + g["pos_stack_3_a1"] * a1 + g["pos_stack_3_a2"] * a2
            self.stack5 = center + g["pos_stack_5_a1"] * a1 + g["pos_stack_5_a2"] * a2
        elif model == 1:
            self.back = center + g["com_to_backbone"] * a1
            self.back_dna1 = self.back
        else:
            self.back = center + g["com_to_backbone_x"] * a1 + g["com_to_backbone_y"] * a2
            self.back_dna1 = center + g["com_to_backbone_dna1"] * a1


def make_displacement(box):
    """jax_md.space.free / space.periodic(box): d(a, b) = a - b (minimum image)."""
    if box is None:
        return lambda a, b: a - b
    side = _t(box)

    def disp(a, b):
        d = a - b
        return torch.remainder(d + side * 0.5, side) - side * 0.5

    return disp


def _dot(a, b):
    return (a * b).sum(-1)


def _norm(a):
    return torch.sqrt((a * a).sum(-1))


# ---------------------------------------------------------------------------------------------
# per-pair terms
# ---------------------------------------------------------------------------------------------


def pair_fene(p, s: Sites, bonded, disp):
    """dna1/fene.py:37-56 + dna1/interactions.py:16-41."""
    i, j = bonded[:, 0], bonded[:, 1]
    r = _norm(disp(s.back[i], s.back[j]))
    eps, r0, delt, fmax, finf = p["eps_backbone"], p["r0_backbone"], p["delta_backbone"], p["fmax"], p["finf"]
    diff = smooth_abs(r - r0)
    xmax = (-eps + torch.sqrt(eps**2 + 4 * fmax**2 * delt**2)) / (2 * fmax)
    fene_xmax = -(eps / 2.0) * torch.log(1.0 - xmax**2 / delt**2)
    long_xmax = (fmax - finf) * xmax * torch.log(xmax) + finf * xmax
    smoothed = (fmax - finf) * xmax * torch.log(diff) + finf * diff - long_xmax + fene_xmax
    use_smooth = diff > xmax
    r_safe = torch.where(use_smooth, r0 + 0.0 * r, r)  # keep log(1-x) finite in the untaken branch
    return torch.where(use_smooth, smoothed, v_fene(r_safe, eps, r0, delt))


def pair_exc_vol_bonded(p, s: Sites, bonded, disp):
    """dna1/bonded_excluded_volume.py:84-114 + dna1/interactions.py:44-83."""
    i, j = bonded[:, 0], bonded[:, 1]
    r_base = _norm(disp(s.base[i], s.base[j]))
    r_back_base = _norm(disp(s.back[i], s.base[j]))
    r_base_back = _norm(disp(s.base[i], s.back[j]))
    e = p["eps_exc"]
    return (
        f3(r_base, p["dr_star_base"], p["dr_c_base"], e, p["sigma_base"], p["b_base"])
        + f3(r_back_base, p["dr_star_back_base"], p["dr_c_back_base"], e, p["sigma_back_base"], p["b_back_base"])
        + f3(r_base_back, p["dr_star_base_back"], p["dr_c_base_back"], e, p["sigma_base_back"], p["b_base_back"])
    )


# base-pair types AT, TA, GC, CG as (nt of member 0, nt of member 1) (mythos/utils/constants.py:13-18)
BP_IDXS = torch.tensor([[0, 3], [3, 0], [2, 1], [1, 2]], dtype=torch.long)


def compute_seq_dep_weight(pseq, nt1, nt2, weights_table, sc):
    """Expected sequence-dependent weight of the pairs (nt1[k], nt2[k]) under a probabilistic sequence:
    mythos/energy/utils.py:45-132 case by case (the reference vmaps a scalar function over the pairs; here the pair
    axis is the leading tensor axis).  ``pseq = (unpaired (n_unpaired, 4), base-pair types (n_bp, 4))``; ``sc`` carries
    ``is_unpaired``, ``idx_to_unpaired_idx``, ``idx_to_bp_idx``.  Index -1 (a nucleotide that is not of the kind a case
    assumes) reads the last row, as in jnp; that case is then discarded by the final selection."""
    up, bp = (torch.as_tensor(a, dtype=torch.float64) for a in pseq)
    if up.shape[0] == 0:
        up = torch.zeros((1, 4), dtype=torch.float64)
    if bp.shape[0] == 0:
        bp = torch.zeros((1, 4), dtype=torch.float64)
    table = torch.as_tensor(weights_table, dtype=torch.float64)
    is_up = torch.as_tensor(sc.is_unpaired).bool()
    iu = torch.as_tensor(sc.idx_to_unpaired_idx, dtype=torch.long)
    ib = torch.as_tensor(sc.idx_to_bp_idx, dtype=torch.long)
    u1, u2 = is_up[nt1], is_up[nt2]
    p1, p2 = up[iu[nt1]], up[iu[nt2]]                      # (P, 4) nucleotide probabilities if unpaired
    b1, w1 = bp[ib[nt1, 0]], ib[nt1, 1]                    # (P, 4) base-pair type probabilities, position inside the pair
    b2, w2 = bp[ib[nt2, 0]], ib[nt2, 1]
    n1_of_t = BP_IDXS[:, w1].T                             # (P, 4): nucleotide of nt1 for each base-pair type
    n2_of_t = BP_IDXS[:, w2].T
    # case 1: both unpaired - kron(p1, p2) . table
    both_unpaired = torch.einsum("pa,pb,ab->p", p1, p2, table)
    # case 2: nt1 unpaired, nt2 in a base pair
    nt1_up = torch.einsum("pa,pt,pat->p", p1, b2, table[:, n2_of_t].permute(1, 0, 2))
    # case 3: nt2 unpaired, nt1 in a base pair
    nt2_up = torch.einsum("pb,pt,ptb->p", p2, b1, table[n1_of_t, :])
    # case 4.I: the two members of one base pair
    same_bp = (b1 * table[n1_of_t, n2_of_t]).sum(-1)
    # case 4.II: members of different base pairs
    diff_bps = torch.einsum("ps,pt,pst->p", b1, b2, table[n1_of_t[:, :, None], n2_of_t[:, None, :]])
    both_paired = torch.where(ib[nt1, 0] == ib[nt2, 0], same_bp, diff_bps)
    return torch.where(u1 & u2, both_unpaired, torch.where(u1, nt1_up, torch.where(u2, nt2_up, both_paired)))


def _seq_weights(p, table_name, seq, i, j):
    """Discrete lookup table[seq_i, seq_j], or the expectation under p["pseq"] (dna1/stacking.py:284-287,
    dna1/hydrogen_bonding.py:330-333)."""
    if p.get("pseq") is not None:
        if p.get("pseq_constraints") is None:
            raise ValueError("pseq_constraints must be provided when pseq is provided.")
        return compute_seq_dep_weight(p["pseq"], i, j, p[table_name], p["pseq_constraints"])
    return p[table_name][seq[i], seq[j]]


def pair_stacking(p, s: Sites, seq, bonded, disp):
    """dna1/stacking.py:192-289; dna2/stacking.py:19-39 (cos(phi) on the dna1 backbone site)."""
    i, j = bonded[:, 0], bonded[:, 1]
    dr_back = disp(s.back_dna1[i], s.back_dna1[j])
    r_back = _norm(dr_back)
    dr_stack = disp(s.stack[i], s.stack[j])
    r_stack = _norm(dr_stack)
    theta4 = torch.acos(clamp(_dot(s.a3[i], s.a3[j])))
    theta5 = PI - torch.acos(clamp(_dot(dr_stack, s.a3[j]) / r_stack))
    theta6 = PI - torch.acos(clamp(_dot(s.a3[i], dr_stack) / r_stack))
    cosphi1 = -_dot(s.a2[i], dr_back) / r_back
    cosphi2 = -_dot(s.a2[j], dr_back) / r_back
    v = (
        f1(
            r_stack,
            p["dr_low_stack"],
            p["dr_high_stack"],
            p["dr_c_low_stack"],
            p["dr_c_high_stack"],
            _t(1.0),
            p["a_stack"],
            p["dr0_stack"],
            p["dr_c_stack"],
            p["b_low_stack"],
            p["b_high_stack"],
        )
        * f4(theta4, p["theta0_stack_4"], p["delta_theta_star_stack_4"], p["delta_theta_stack_4_c"], p["a_stack_4"], p["b_stack_4"])
        * f4(theta5, p["theta0_stack_5"], p["delta_theta_star_stack_5"], p["delta_theta_stack_5_c"], p["a_stack_5"], p["b_stack_5"])
        * f4(theta6, p["theta0_stack_6"], p["delta_theta_star_stack_6"], p["delta_theta_stack_6_c"], p["a_stack_6"], p["b_stack_6"])
        * f5(-cosphi1, p["neg_cos_phi1_star_stack"], p["neg_cos_phi1_c_stack"], p["a_stack_1"], p["b_neg_cos_phi1_stack"])
        * f5(-cosphi2, p["neg_cos_phi2_star_stack"], p["neg_cos_phi2_c_stack"], p["a_stack_2"], p["b_neg_cos_phi2_stack"])
    )
    return _seq_weights(p, "eps_stack", seq, i, j) * v


def pair_stacking_rna2(p, s: Sites, seq, bonded, disp):
    """rna2/stacking.py:186-292 + rna2/interactions.py:15-135: the stacking sites are the 5' site of nn_i and the 3' site
    of nn_j, there is no theta4, and theta9 / theta10 measure the backbone vector against the p3 vector of nn_j and
    the p5 vector of nn_i."""
    i, j = bonded[:, 0], bonded[:, 1]
    dr_stack = disp(s.stack5[i], s.stack3[j])
    r_stack = _norm(dr_stack)
    theta5 = PI - torch.acos(clamp(_dot(dr_stack, s.a3[j]) / r_stack))
    theta6 = PI - torch.acos(clamp(_dot(s.a3[i], dr_stack) / r_stack))
    dr_back = disp(s.back[i], s.back[j])
    r_back = _norm(dr_back)
    theta9 = torch.acos(clamp(_dot(-s.p3[j], dr_back) / r_back))
    theta10 = torch.acos(clamp(_dot(-s.p5[i], dr_back) / r_back))
    cosphi1 = -_dot(s.a2[i], dr_back) / r_back
    cosphi2 = -_dot(s.a2[j], dr_back) / r_back
    v = f1(r_stack, p["dr_low_stack"], p["dr_high_stack"], p["dr_c_low_stack"], p["dr_c_high_stack"], _t(1.0), p["a_stack"],
           p["dr0_stack"], p["dr_c_stack"], p["b_low_stack"], p["b_high_stack"])
    for k, th in ((5, theta5), (6, theta6), (9, theta9), (10, theta10)):
        v = v * f4(th, p[f"theta0_stack_{k}"], p[f"delta_theta_star_stack_{k}"], p[f"delta_theta_stack_{k}_c"], p[f"a_stack_{k}"],
                   p[f"b_stack_{k}"])
    v = v * f5(-cosphi1, p["neg_cos_phi1_star_stack"], p["neg_cos_phi1_c_stack"], p["a_stack_1"], p["b_neg_cos_phi1_stack"])
    v = v * f5(-cosphi2, p["neg_cos_phi2_star_stack"], p["neg_cos_phi2_c_stack"], p["a_stack_2"], p["b_neg_cos_phi2_stack"])
    return _seq_weights(p, "eps_stack", seq, i, j) * v


def pair_exc_vol_unbonded(p, s: Sites, pairs, disp):
    """dna1/unbonded_excluded_volume.py:105-146 + dna1/interactions.py:86-135."""
    i, j = pairs[:, 0], pairs[:, 1]
    r_base = _norm(disp(s.base[j], s.base[i]))
    r_back = _norm(disp(s.back[j], s.back[i]))
    r_back_base = _norm(disp(s.back[i], s.base[j]))
    r_base_back = _norm(disp(s.base[i], s.back[j]))
    e = p["eps_exc"]
    return (
        f3(r_back, p["dr_star_backbone"], p["dr_c_backbone"], e, p["sigma_backbone"], p["b_backbone"])
        + f3(r_base, p["dr_star_base"], p["dr_c_base"], e, p["sigma_base"], p["b_base"])
        + f3(r_back_base, p["dr_star_back_base"], p["dr_c_back_base"], e, p["sigma_back_base"], p["b_back_base"])
        + f3(r_base_back, p["dr_star_base_back"], p["dr_c_base_back"], e, p["sigma_base_back"], p["b_base_back"])
    )


def _hb_angles(s: Sites, pairs, disp):
    """Shared geometry of H-bond and cross-stacking: dna1/hydrogen_bonding.py:244-258,
    dna1/cross_stacking.py:204-218."""
    i, j = pairs[:, 0], pairs[:, 1]
    dr = disp(s.base[j], s.base[i])
    r = _norm(dr)
    t1 = torch.acos(clamp(_dot(-s.a1[i], s.a1[j])))
    t2 = torch.acos(clamp(_dot(-s.a1[j], dr) / r))
    t3 = torch.acos(clamp(_dot(s.a1[i], dr) / r))
    t4 = torch.acos(clamp(_dot(s.a3[i], s.a3[j])))
    t7 = torch.acos(clamp(_dot(-s.a3[j], dr) / r))
    t8 = PI - torch.acos(clamp(_dot(s.a3[i], dr) / r))
    return r, t1, t2, t3, t4, t7, t8


def pair_hydrogen_bonding(p, s: Sites, seq, pairs, disp):
    """dna1/hydrogen_bonding.py:232-335 + dna1/interactions.py:513-640."""
    i, j = pairs[:, 0], pairs[:, 1]
    r, t1, t2, t3, t4, t7, t8 = _hb_angles(s, pairs, disp)
    v = f1(
        r,
        p["dr_low_hb"],
        p["dr_high_hb"],
        p["dr_c_low_hb"],
        p["dr_c_high_hb"],
        _t(1.0),
        p["a_hb"],
        p["dr0_hb"],
        p["dr_c_hb"],
        p["b_low_hb"],
        p["b_high_hb"],
    )
    for k, th in ((1, t1), (2, t2), (3, t3), (4, t4), (7, t7), (8, t8)):
        v = v * f4(th, p[f"theta0_hb_{k}"], p[f"delta_theta_star_hb_{k}"], p[f"delta_theta_hb_{k}_c"], p[f"a_hb_{k}"], p[f"b_hb_{k}"])
    return _seq_weights(p, "eps_hb_weights", seq, i, j) * v


def pair_cross_stacking(p, s: Sites, pairs, disp):
    """dna1/cross_stacking.py:192-266 + dna1/interactions.py:253-385."""
    r, t1, t2, t3, t4, t7, t8 = _hb_angles(s, pairs, disp)

    def g(k, th):
        return f4(th, p[f"theta0_cross_{k}"], p[f"delta_theta_star_cross_{k}"], p[f"delta_theta_cross_{k}_c"], p[f"a_cross_{k}"], p[f"b_cross_{k}"])

    return (
        f2(
            r,
            p["dr_low_cross"],
            p["dr_high_cross"],
            p["dr_c_low_cross"],
            p["dr_c_high_cross"],
            p["k_cross"],
            p["r0_cross"],
            p["dr_c_cross"],
            p["b_low_cross"],
            p["b_high_cross"],
        )
        * g(1, t1)
        * g(2, t2)
        * g(3, t3)
        * ((g(4, t4) + g(4, PI - t4)) if "a_cross_4" in p else 1.0)  # no theta4 factor in oxRNA2 (rna2/interactions.py:238-256)
        * (g(7, t7) + g(7, PI - t7))
        * (g(8, t8) + g(8, PI - t8))
    )


def pair_coaxial(p, s: Sites, pairs, disp, model: int):
    """dna1/coaxial_stacking.py:181-260 + dna1/interactions.py:388-510;
    dna2/coaxial_stacking.py:138-201 + dna2/interactions.py:31-136."""
    i, j = pairs[:, 0], pairs[:, 1]
    dr_stack = disp(s.stack[j], s.stack[i])
    r_stack = _norm(dr_stack)
    n = dr_stack / r_stack[:, None]
    t4 = torch.acos(clamp(_dot(s.a3[i], s.a3[j])))
    t1 = torch.acos(clamp(_dot(-s.a1[i], s.a1[j])))
    t5 = torch.acos(clamp(_dot(s.a3[i], n)))
    t6 = torch.acos(clamp(_dot(-s.a3[j], n)))

    def g(k, th):
        return f4(th, p[f"theta0_coax_{k}"], p[f"delta_theta_star_coax_{k}"], p[f"delta_theta_coax_{k}_c"], p[f"a_coax_{k}"], p[f"b_coax_{k}"])

    rad = f2(
        r_stack,
        p["dr_low_coax"],
        p["dr_high_coax"],
        p["dr_c_low_coax"],
        p["dr_c_high_coax"],
        p["k_coax"],
        p["dr0_coax"],
        p["dr_c_coax"],
        p["b_low_coax"],
        p["b_high_coax"],
    )
    common = rad * g(4, t4) * (g(5, t5) + g(5, PI - t5)) * (g(6, t6) + g(6, PI - t6))
    if model == 1:
        dr_back = disp(s.back[j], s.back[i])
        nb = dr_back / _norm(dr_back)[:, None]
        cosphi3 = _dot(n, torch.linalg.cross(nb, s.a1[j]))
        cosphi4 = _dot(n, torch.linalg.cross(nb, s.a1[i]))
        return (
            common
            * (g(1, t1) + g(1, 2 * PI - t1))
            * f5(cosphi3, p["cos_phi3_star_coax"], p["cos_phi3_c_coax"], p["a_coax_3p"], p["b_cos_phi3_coax"])
            * f5(cosphi4, p["cos_phi4_star_coax"], p["cos_phi4_c_coax"], p["a_coax_4p"], p["b_cos_phi4_coax"])
        )
    return common * (g(1, t1) + f6(t1, p["a_coax_1_f6"], p["b_coax_1_f6"]))


def pair_debye(p, s: Sites, is_end, pairs, disp):
    """dna2/debye.py:82-110 + dna2/interactions.py:15-28."""
    i, j = pairs[:, 0], pairs[:, 1]
    r = _norm(disp(s.back[j], s.back[i]))
    full = torch.exp(r * -p["kappa"]) * (p["prefactor"] / r)
    smooth = p["smoothing_coeff"] * (r - p["r_cut"]) ** 2
    e = torch.where(r < p["r_high"], full, smooth)
    e = torch.where(r < p["r_cut"], e, torch.zeros_like(e))
    if p["half_charged_ends"]:
        half = torch.full_like(r, 0.5)
        one = torch.ones_like(r)
        e = e * torch.where(is_end[i] != 0, half, one) * torch.where(is_end[j] != 0, half, one)
    return e


# ---------------------------------------------------------------------------------------------
# whole-system evaluation
# ---------------------------------------------------------------------------------------------


def pair_terms(model, P, center, quat, seq, is_end, bonded, unbonded, box=None, axes=None):
    """Per-pair energies of one frame.

    Returns ``(bonded_terms, unbonded_terms)``: dicts term-name -> (B,) / (P,) tensors.
    ``axes=(a1, a3)`` bypasses the quaternion (golden frames give axes directly).
    """
    if axes is None:
        a1, a2, a3 = quat_to_axes(quat)
    else:
        a1, a3 = axes
        a2 = torch.linalg.cross(a3, a1)
    s = Sites(model, P["geometry"], center, a1, a2, a3)
    disp = make_displacement(box)
    bt = {
        "fene": pair_fene(P["fene"], s, bonded, disp),
        "bonded_excluded_volume": pair_exc_vol_bonded(P["bonded_excluded_volume"], s, bonded, disp),
        "stacking": (pair_stacking_rna2 if model == 3 else pair_stacking)(P["stacking"], s, seq, bonded, disp),
    }
    ut = {
        "unbonded_excluded_volume": pair_exc_vol_unbonded(P["unbonded_excluded_volume"], s, unbonded, disp),
        "hydrogen_bonding": pair_hydrogen_bonding(P["hydrogen_bonding"], s, seq, unbonded, disp),
        "cross_stacking": pair_cross_stacking(P["cross_stacking"], s, unbonded, disp),
        "coaxial_stacking": pair_coaxial(P["coaxial_stacking"], s, unbonded, disp, 1 if model == 3 else model),
    }
    if model in (2, 3):
        ut["debye"] = pair_debye(P["debye"], s, is_end, unbonded, disp)
    return bt, ut


# ---------------------------------------------------------------------------------------------
# oxNA (model 4): hybrid DNA / RNA systems, mythos/energy/na1/*.py
# ---------------------------------------------------------------------------------------------
NA1_UNBONDED = ("unbonded_excluded_volume", "hydrogen_bonding", "cross_stacking", "coaxial_stacking", "debye")


def init_all_na1(cfg_dna: dict, cfg_rna: dict, cfg_drh: dict, kt, salt_conc=0.5, half_charged_ends=False) -> dict:
    """Three parameter sets, as every na1 ``*Configuration.init_params`` builds them (e.g. na1/hydrogen_bonding.py:215-307):
    ``dna`` (oxDNA2 sections), ``rna`` (oxRNA2 sections) and ``drh`` (the DNA-RNA hybrid numbers of
    mythos/input/na1/default_energy.toml, unbonded terms only, all in their oxDNA1 functional form)."""
    return {
        "dna": init_all(2, cfg_dna, kt, salt_conc, half_charged_ends),
        "rna": init_all(3, cfg_rna, kt, salt_conc, half_charged_ends),
        "drh": {
            "unbonded_excluded_volume": init_exc_vol(cfg_drh["unbonded_excluded_volume"], with_backbone=True),
            "hydrogen_bonding": init_hydrogen_bonding(cfg_drh["hydrogen_bonding"]),
            "cross_stacking": init_cross_stacking(cfg_drh["cross_stacking"]),
            "coaxial_stacking": init_coaxial(cfg_drh["coaxial_stacking"], 1),
            "debye": init_debye({**cfg_drh["debye"], "kt": kt, "salt_conc": salt_conc, "half_charged_ends": half_charged_ends}),
        },
    }


class _MixedSites:
    """Sites of every nucleotide by its OWN type: ``pairwise_energies(nucleotide.dna, nucleotide.rna, ...)`` of the
    hybrid branches (na1/hydrogen_bonding.py:336-348 and the other unbonded terms) reads nn_i from the oxDNA2 sites and
    nn_j from the oxRNA2 sites; for a DNA-DNA or RNA-RNA pair the same table gives both from one model."""

    def __init__(self, dna: Sites, rna: Sites, is_rna):
        m = is_rna[:, None]
        self.a1, self.a2, self.a3, self.center = dna.a1, dna.a2, dna.a3, dna.center
        for name in ("stack", "base", "back"):
            setattr(self, name, torch.where(m, getattr(rna, name), getattr(dna, name)))


def pair_terms_na1(P, center, quat, seq, nt_is_rna, is_end, bonded, unbonded, box=None, axes=None):
    """Per-pair energies of one frame of a hybrid system.  Bonded pairs: both RNA -> oxRNA2 forms on the oxRNA2 sites,
    otherwise the oxDNA2 forms on the oxDNA2 sites (na1/fene.py:89-106, bonded_excluded_volume.py:96-116,
    stacking.py:193-217).  Unbonded pairs: RNA-RNA -> oxRNA2, DNA-DNA -> oxDNA2, DNA-RNA and RNA-DNA -> the ``drh``
    parameters in the oxDNA1 forms with each nucleotide's own sites (na1/unbonded_excluded_volume.py:140-174,
    hydrogen_bonding.py:314-362, cross_stacking.py:262-300, coaxial_stacking.py:249-287, debye.py:103-141)."""
    if axes is None:
        a1, a2, a3 = quat_to_axes(quat)
    else:
        a1, a3 = axes
        a2 = torch.linalg.cross(a3, a1)
    sd = Sites(2, P["dna"]["geometry"], center, a1, a2, a3)
    sr = Sites(3, P["rna"]["geometry"], center, a1, a2, a3)
    rna = nt_is_rna.bool()
    disp = make_displacement(box)
    # every branch is evaluated on ITS pairs only (the reference evaluates all branches on all pairs and selects with
    # jnp.where; the values are the same, and no unselected branch can put a nan into the gradients)
    def scatter(total, parts):
        out = torch.zeros(total, dtype=center.dtype)
        for idx, val in parts:
            out = out.index_put((idx,), val)
        return out

    rna_bond = rna[bonded[:, 0]] & rna[bonded[:, 1]]
    ir, idn = torch.nonzero(rna_bond)[:, 0], torch.nonzero(~rna_bond)[:, 0]
    br, bd = bonded[ir], bonded[idn]
    nb = bonded.shape[0]
    bt = {
        "fene": scatter(nb, [(ir, pair_fene(P["rna"]["fene"], sr, br, disp)), (idn, pair_fene(P["dna"]["fene"], sd, bd, disp))]),
        "bonded_excluded_volume": scatter(nb, [(ir, pair_exc_vol_bonded(P["rna"]["bonded_excluded_volume"], sr, br, disp)),
                                               (idn, pair_exc_vol_bonded(P["dna"]["bonded_excluded_volume"], sd, bd, disp))]),
        "stacking": scatter(nb, [(ir, pair_stacking_rna2(P["rna"]["stacking"], sr, seq, br, disp)),
                                 (idn, pair_stacking(P["dna"]["stacking"], sd, seq, bd, disp))]),
    }
    ui, uj = unbonded[:, 0], unbonded[:, 1]
    both_rna = rna[ui] & rna[uj]
    both_dna = ~rna[ui] & ~rna[uj]
    sm = _MixedSites(sd, sr, rna)
    kinds = (("rna", torch.nonzero(both_rna)[:, 0]), ("dna", torch.nonzero(both_dna)[:, 0]),
             ("drh", torch.nonzero(~both_rna & ~both_dna)[:, 0]))
    nu = unbonded.shape[0]

    def three(name, fn):
        return scatter(nu, [(idx, fn(P[k][name], unbonded[idx], k)) for k, idx in kinds])

    ut = {
        "unbonded_excluded_volume": three("unbonded_excluded_volume", lambda p, u, k: pair_exc_vol_unbonded(p, sm, u, disp)),
        "hydrogen_bonding": three("hydrogen_bonding", lambda p, u, k: pair_hydrogen_bonding(p, sm, seq, u, disp)),
        # cross-stacking: oxDNA form (theta4 factor) for DNA-DNA and the hybrids, oxRNA2 form for RNA-RNA - the parameter
        # set decides (the oxRNA2 section has no theta4 block)
        "cross_stacking": three("cross_stacking", lambda p, u, k: pair_cross_stacking(p, sm, u, disp)),
        # coaxial stacking: the oxDNA2 form (f6) for DNA-DNA, the oxDNA1 form (f5 of cos phi3 / phi4) for RNA-RNA and hybrids
        "coaxial_stacking": three("coaxial_stacking", lambda p, u, k: pair_coaxial(p, sm, u, disp, 2 if k == "dna" else 1)),
        "debye": three("debye", lambda p, u, k: pair_debye(p, sm, is_end, u, disp)),
    }
    return bt, ut


def energy_terms_na1(P, center, quat, seq, nt_is_rna, is_end, bonded, unbonded, box=None, axes=None):
    """(8,) term energies of a hybrid system, in the oxDNA2 column order."""
    bt, ut = pair_terms_na1(P, center, quat, seq, nt_is_rna, is_end, bonded, unbonded, box, axes)
    both = {**bt, **ut}
    return torch.stack([both[n].sum() for n in TERMS_DNA2])


def energy_and_grads_na1(P, center, quat, seq, nt_is_rna, is_end, bonded, unbonded, box=None):
    c = center.detach().clone().requires_grad_(True)
    q = quat.detach().clone().requires_grad_(True)
    u = energy_terms_na1(P, c, q, seq, nt_is_rna, is_end, bonded, unbonded, box).sum()
    gc, gq = torch.autograd.grad(u, (c, q))
    return u.detach(), gc, gq


def energy_terms(model, P, center, quat, seq, is_end, bonded, unbonded, box=None, axes=None):
    """(n_terms,) tensor in the reference's term order (dna1: 7 terms, dna2: 8).

    Equivalent of ``ComposedEnergyFunction.compute_terms`` (mythos/energy/base.py:312-314).
    """
    bt, ut = pair_terms(model, P, center, quat, seq, is_end, bonded, unbonded, box, axes)
    names = TERMS_DNA1 if model == 1 else TERMS_DNA2  # oxRNA2 reports the eight oxDNA2 columns
    both = {**bt, **ut}
    return torch.stack([both[n].sum() for n in names])


def energy(model, P, center, quat, seq, is_end, bonded, unbonded, box=None):
    """Total potential energy of one frame (``ComposedEnergyFunction.__call__``, base.py:316-319)."""
    return energy_terms(model, P, center, quat, seq, is_end, bonded, unbonded, box).sum()


def energy_and_grads(model, P, center, quat, seq, is_end, bonded, unbonded, box=None):
    """U, dU/dcenter (N,3), dU/dquat (N,4) by autograd - the stand-in for jax.grad."""
    c = center.detach().clone().requires_grad_(True)
    q = quat.detach().clone().requires_grad_(True)
    u = energy(model, P, c, q, seq, is_end, bonded, unbonded, box)
    gc, gq = torch.autograd.grad(u, (c, q))
    return u.detach(), gc, gq


def quat_grad_to_body_torque(quat, dU_dq):
    """Body-frame torque from the quaternion gradient: tau_k = -1/2 (P_k q) . dU/dq.

    P_1 q = (-q1, q0, q3, -q2), P_2 q = (-q2, -q3, q0, q1), P_3 q = (-q3, q2, -q1, q0)
    (Miller et al. 2002 NO_SQUISH permutations; body axes 1,2,3 = a1,a2,a3).
    """
    q0, q1, q2, q3 = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    p1 = torch.stack([-q1, q0, q3, -q2], dim=-1)
    p2 = torch.stack([-q2, -q3, q0, q1], dim=-1)
    p3 = torch.stack([-q3, q2, -q1, q0], dim=-1)
    return -0.5 * torch.stack([_dot(p1, dU_dq), _dot(p2, dU_dq), _dot(p3, dU_dq)], dim=-1)
