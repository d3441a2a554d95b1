"""Independent parameters -> the flat parameter vector of the HIP kernels (torch fp64, differentiable).

The reference derives its "dependent" constants inside every ``*Configuration.init_params``
(mythos/energy/dna1/stacking.py:120-183, hydrogen_bonding.py:148-223, cross_stacking.py:110-183,
coaxial_stacking.py:106-172, dna2/coaxial_stacking.py:79-130, excluded volume :56-75 / :67-96,
dna2/debye.py:47-64) from sympy-expanded closed forms (base_smoothing_functions.py:13-142).
Every one of those constants is the solution of the same problem: continue a core function
V(x) beyond a breakpoint x* by a parabola b (x_c - x)^2 with matching value and slope, i.e.

    x_c = x* - 2 V(x*) / V'(x*),        b = V'(x*)^2 / (4 V(x*)).

That form is used here (``c1_match``); it agrees with the reference's expressions to rounding
(tests/test_flat_params.py) and keeps the whole map differentiable, which is how the chain
rule  dU/dtheta = (d flat / d theta)^T dU/dflat  is applied on the host.
"""

from __future__ import annotations

import math

import numpy as np
import torch

F64 = torch.float64


def _t(x) -> torch.Tensor:
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=F64)


class _Torch:
    """The arithmetic of ``derive_flat`` on 0-d torch tensors: differentiable (the chain rule of dU/dtheta runs through it)."""

    exp, sqrt, log = torch.exp, torch.sqrt, torch.log
    t = staticmethod(_t)

    @staticmethod
    def ones(n):
        return torch.ones(n, dtype=F64)


class _Numpy:
    """The same arithmetic on numpy float64 scalars, for callers that need numbers, not a graph (a simulator run, ``map``
    without gradients): ~300 scalar operations cost 0.85 ms as torch tensors and 0.06 ms as numpy scalars, and the packed
    vector another 0.3 ms of ``torch.stack`` (round 4: what was left of HipMDSimulator.run's host time)."""

    exp, sqrt, log = np.exp, np.sqrt, np.log

    @staticmethod
    def t(x):
        if isinstance(x, torch.Tensor):
            x = x.detach().cpu().numpy()
        a = np.asarray(x, dtype=np.float64)
        return np.float64(a) if a.ndim == 0 else a

    @staticmethod
    def ones(n):
        return np.ones(n, dtype=np.float64)


def _wants_graph(*values) -> bool:
    for v in values:
        if isinstance(v, torch.Tensor):
            if v.requires_grad:
                return True
        elif isinstance(v, dict):
            if _wants_graph(*v.values()):
                return True
        elif isinstance(v, (tuple, list)):
            if _wants_graph(*v):
                return True
    return False


def c1_match(x, v, dv):
    """(b, x_c) of the parabola b (x_c - x)^2 matching value v and slope dv at x."""
    return dv * dv / (4.0 * v), x - 2.0 * v / dv


def _f1_block(prefix, r_low, r_high, a, r0, r_c, out, xp=_Torch):
    """f1 = Morse(r) - Morse(r_c), eps = 1 (dna1/base_functions.py:13-37)."""
    shift = (1.0 - xp.exp(-a * (r_c - r0))) ** 2

    def v(x):
        e = xp.exp(-a * (x - r0))
        return (1.0 - e) ** 2 - shift, 2.0 * a * e * (1.0 - e)

    b_low, rc_low = c1_match(r_low, *v(r_low))
    b_high, rc_high = c1_match(r_high, *v(r_high))
    out.update(
        {
            f"{prefix}_RLOW": r_low,
            f"{prefix}_RHIGH": r_high,
            f"{prefix}_RCLOW": rc_low,
            f"{prefix}_RCHIGH": rc_high,
            f"{prefix}_A": a,
            f"{prefix}_R0": r0,
            f"{prefix}_RC": r_c,
            f"{prefix}_BLOW": b_low,
            f"{prefix}_BHIGH": b_high,
            f"{prefix}_SHIFT": shift,
        }
    )


def _f2_block(prefix, r_low, r_high, k, r0, r_c, out):
    """f2 = k [ (r - r0)^2/2 - (r_c - r0)^2/2 ] (dna1/base_functions.py:40-63); k factors out of b."""
    shift = 0.5 * (r_c - r0) ** 2

    def v(x):
        return 0.5 * (x - r0) ** 2 - shift, x - r0

    b_low, rc_low = c1_match(r_low, *v(r_low))
    b_high, rc_high = c1_match(r_high, *v(r_high))
    out.update(
        {
            f"{prefix}_RLOW": r_low,
            f"{prefix}_RHIGH": r_high,
            f"{prefix}_RCLOW": rc_low,
            f"{prefix}_RCHIGH": rc_high,
            f"{prefix}_K": k,
            f"{prefix}_R0": r0,
            f"{prefix}_RC": r_c,
            f"{prefix}_BLOW": b_low,
            f"{prefix}_BHIGH": b_high,
            f"{prefix}_SHIFT": shift,
        }
    )


def _f3_block(prefix, r_star, sigma, out):
    """f3 = 4 eps [ (s/r)^12 - (s/r)^6 ] (dna1/base_functions.py:66-79); eps factors out of b."""
    s6 = (sigma / r_star) ** 6
    v = 4.0 * (s6 * s6 - s6)
    dv = -24.0 * (2.0 * s6 * s6 - s6) / r_star
    b, rc = c1_match(r_star, v, dv)
    out.update({f"{prefix}_RSTAR": r_star, f"{prefix}_SIGMA": sigma, f"{prefix}_B": b, f"{prefix}_RC": rc})


def _f4_block(prefix, theta0, dts, a, out):
    """f4 = 1 - a (t - t0)^2 (dna1/base_functions.py:82-107), matched at t0 + dts (symmetric)."""
    b, tc = c1_match(theta0 + dts, 1.0 - a * dts * dts, -2.0 * a * dts)
    out.update(
        {f"{prefix}_T0": theta0, f"{prefix}_TS": dts, f"{prefix}_TC": tc - theta0, f"{prefix}_A": a, f"{prefix}_B": b}
    )


def _f5_block(prefix, x_star, a, out):
    """f5 = 1 - a x^2 on (x*, 0) (dna1/base_functions.py:110-129), matched at x* < 0."""
    b, xc = c1_match(x_star, 1.0 - a * x_star * x_star, -2.0 * a * x_star)
    out.update({f"{prefix}_XS": x_star, f"{prefix}_XC": xc, f"{prefix}_A": a, f"{prefix}_B": b})


STACK_WEIGHTS_SA = torch.ones(4, 4, dtype=F64)
HB_WEIGHTS_SA = torch.tensor(
    [[0.0, 0.0, 0.0, 1.0], [0.0, 0.0, 1.0, 0.0], [0.0, 1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]], dtype=F64
)


TERM_WEIGHT_NAMES = ("TW_FENE", "TW_BEXC", "TW_STCK", "TW_NEXC", "TW_HB", "TW_CRST", "TW_CXST", "TW_DH")


def derive_flat(
    model: int, sections: dict, *, kt, salt_conc=0.5, half_charged_ends=True, term_weights=None, numbers_ok: bool = False
) -> dict[str, torch.Tensor]:
    """Name -> fp64 tensor for every entry of the kernels' flat parameter vector.

    ``sections`` is TOML-shaped: {"fene": {...}, "stacking": {...}, ...} with the reference's
    parameter names; values may be floats or (requires_grad) tensors.  ``stacking`` may hold
    ``ss_stack_weights`` (4,4) and ``hydrogen_bonding`` ``ss_hb_weights`` (4,4).  ``term_weights`` (8,)
    scale the gradients of the eight terms (ComposedEnergyFunction.weights); energies stay unweighted.
    ``numbers_ok``: the caller only packs the result (``pack_flat``); when no input requires a gradient the derivation then
    runs on numpy scalars and the values come back as numpy float64 (same formulas, same rounding to the last digits).
    """
    xp = _Numpy if (numbers_ok and not _wants_graph(sections, kt, salt_conc, term_weights)) else _Torch
    _t = xp.t  # noqa: F811 - the conversion of this backend
    S = {sec: {k: (_t(v) if v is not None else None) for k, v in d.items()} for sec, d in sections.items()}
    kt = _t(kt)
    out: dict[str, torch.Tensor] = {}
    zero = _t(0.0)

    g = S["geometry"]
    rna_only = ("GEO_STACK3_A1", "GEO_STACK3_A2", "GEO_STACK5_A1", "GEO_STACK5_A2", "GEO_P3_X", "GEO_P3_Y", "GEO_P3_Z",
                "GEO_P5_X", "GEO_P5_Y", "GEO_P5_Z")
    if model == 3:
        # oxRNA2 (rna2/nucleotide.py:52-63): the backbone site is c + pos_back_a1 a1 + pos_back_a3 a3 (the kernels'
        # model-3 instantiation puts the second coefficient on a3); stacking runs between separate 3' / 5' sites.
        out["GEO_STACK"] = g["pos_stack"]
        out["GEO_BASE"] = g["pos_base"]
        out["GEO_BACK_A1"] = g["pos_back_a1"]
        out["GEO_BACK_A2"] = g["pos_back_a3"]
        out["GEO_BACK_DNA1"] = g["pos_back_a1"]  # unused by model 3
        out.update(GEO_STACK3_A1=g["pos_stack_3_a1"], GEO_STACK3_A2=g["pos_stack_3_a2"], GEO_STACK5_A1=g["pos_stack_5_a1"],
                   GEO_STACK5_A2=g["pos_stack_5_a2"], GEO_P3_X=g["p3_x"], GEO_P3_Y=g["p3_y"], GEO_P3_Z=g["p3_z"],
                   GEO_P5_X=g["p5_x"], GEO_P5_Y=g["p5_y"], GEO_P5_Z=g["p5_z"])
    else:
        out["GEO_STACK"] = g["com_to_stacking"]
        out["GEO_BASE"] = g["com_to_hb"]
        out.update({k: zero for k in rna_only})
    if model == 3:
        pass
    elif model == 1:
        out["GEO_BACK_A1"] = g["com_to_backbone"]
        out["GEO_BACK_A2"] = zero
        out["GEO_BACK_DNA1"] = g["com_to_backbone"]
    else:
        out["GEO_BACK_A1"] = g["com_to_backbone_x"]
        out["GEO_BACK_A2"] = g["com_to_backbone_y"]
        out["GEO_BACK_DNA1"] = g["com_to_backbone_dna1"]

    # FENE (dna1/interactions.py:16-41)
    f = S["fene"]
    eps, r0, delta, fmax, finf = f["eps_backbone"], f["r0_backbone"], f["delta_backbone"], f["fmax"], f["finf"]
    xmax = (-eps + xp.sqrt(eps**2 + 4 * fmax**2 * delta**2)) / (2 * fmax)
    fene_xmax = -(eps / 2.0) * xp.log(1.0 - xmax**2 / delta**2)
    long_xmax = (fmax - finf) * xmax * xp.log(xmax) + finf * xmax
    out.update(
        FENE_EPS=eps, FENE_R0=r0, FENE_DELTA=delta, FENE_FMAX=fmax, FENE_FINF=finf, FENE_XMAX=xmax,
        FENE_CONST=fene_xmax - long_xmax,
    )

    # excluded volume
    b = S["bonded_excluded_volume"]
    out["BEXC_EPS"] = b["eps_exc"]
    _f3_block("BEXC_BASE", b["dr_star_base"], b["sigma_base"], out)
    _f3_block("BEXC_BACK_BASE", b["dr_star_back_base"], b["sigma_back_base"], out)
    _f3_block("BEXC_BASE_BACK", b["dr_star_base_back"], b["sigma_base_back"], out)
    u = S["unbonded_excluded_volume"]
    out["NEXC_EPS"] = u["eps_exc"]
    _f3_block("NEXC_BASE", u["dr_star_base"], u["sigma_base"], out)
    _f3_block("NEXC_BACK_BASE", u["dr_star_back_base"], u["sigma_back_base"], out)
    _f3_block("NEXC_BASE_BACK", u["dr_star_base_back"], u["sigma_base_back"], out)
    _f3_block("NEXC_BACKBONE", u["dr_star_backbone"], u["sigma_backbone"], out)

    # stacking (dna1/stacking.py:120-183)
    st = S["stacking"]
    st_kt = st.get("kt", kt)
    st_kt = kt if st_kt is None else st_kt
    _f1_block("STCK", st["dr_low_stack"], st["dr_high_stack"], st["a_stack"], st["dr0_stack"], st["dr_c_stack"], out, xp)
    # oxDNA: theta 4, 5, 6; oxRNA2: theta 5, 6, 9, 10 (rna2/stacking.py:60-176).  The blocks a model does not have are
    # filled with a well-formed unused modulation (the kernels never read them).
    for k in (4, 5, 6, 9, 10):
        if f"a_stack_{k}" in st:
            _f4_block(f"STCK_TH{k}", st[f"theta0_stack_{k}"], st[f"delta_theta_star_stack_{k}"], st[f"a_stack_{k}"], out)
        else:
            _f4_block(f"STCK_TH{k}", zero, _t(0.5), _t(1.0), out)
    for k in (1, 2):
        _f5_block(f"STCK_PHI{k}", st[f"neg_cos_phi{k}_star_stack"], st[f"a_stack_{k}"], out)
    if st.get("ss_stack_weights") is None:
        eps_stack = (st["eps_stack_base"] + st["eps_stack_kt_coeff"] * st_kt) * _t(STACK_WEIGHTS_SA)
    else:
        eps_stack = st["ss_stack_weights"] * (1.0 - st["eps_stack_kt_coeff"] + (st_kt * 9.0 * st["eps_stack_kt_coeff"]))

    # hydrogen bonding (dna1/hydrogen_bonding.py:148-223)
    hb = S["hydrogen_bonding"]
    _f1_block("HYDR", hb["dr_low_hb"], hb["dr_high_hb"], hb["a_hb"], hb["dr0_hb"], hb["dr_c_hb"], out, xp)
    for k in (1, 2, 3, 4, 7, 8):
        _f4_block(f"HYDR_TH{k}", hb[f"theta0_hb_{k}"], hb[f"delta_theta_star_hb_{k}"], hb[f"a_hb_{k}"], out)
    eps_hb = _t(HB_WEIGHTS_SA) * hb["eps_hb"] if hb.get("ss_hb_weights") is None else hb["ss_hb_weights"]

    # cross stacking (dna1/cross_stacking.py:110-183)
    cr = S["cross_stacking"]
    _f2_block("CRST", cr["dr_low_cross"], cr["dr_high_cross"], cr["k_cross"], cr["r0_cross"], cr["dr_c_cross"], out)
    for k in (1, 2, 3, 4, 7, 8):
        if f"a_cross_{k}" in cr:
            _f4_block(f"CRST_TH{k}", cr[f"theta0_cross_{k}"], cr[f"delta_theta_star_cross_{k}"], cr[f"a_cross_{k}"], out)
        else:  # oxRNA2 has no theta4 factor in cross-stacking (rna2/cross_stacking.py:97-147)
            _f4_block(f"CRST_TH{k}", zero, _t(0.5), _t(1.0), out)

    # coaxial stacking (dna1/coaxial_stacking.py:106-172, dna2/coaxial_stacking.py:79-130)
    cx = S["coaxial_stacking"]
    _f2_block("CXST", cx["dr_low_coax"], cx["dr_high_coax"], cx["k_coax"], cx["dr0_coax"], cx["dr_c_coax"], out)
    for k in (4, 1, 5, 6):
        _f4_block(f"CXST_TH{k}", cx[f"theta0_coax_{k}"], cx[f"delta_theta_star_coax_{k}"], cx[f"a_coax_{k}"], out)
    if model in (1, 3):  # oxRNA2 keeps the oxDNA1 form of the coaxial term (rna2/tests/test_integration.py:258-287)
        _f5_block("CXST_PHI3", cx["cos_phi3_star_coax"], cx["a_coax_3p"], out)
        _f5_block("CXST_PHI4", cx["cos_phi4_star_coax"], cx["a_coax_4p"], out)
        out["CXST_F6_A"] = zero
        out["CXST_F6_B"] = _t(4.0)  # theta >= b never true (theta <= pi): f6 == 0
    else:
        for k in (3, 4):
            out.update({f"CXST_PHI{k}_XS": _t(-0.5), f"CXST_PHI{k}_XC": _t(-1.0), f"CXST_PHI{k}_A": zero, f"CXST_PHI{k}_B": zero})
        out["CXST_F6_A"] = cx["a_coax_1_f6"]
        out["CXST_F6_B"] = cx["b_coax_1_f6"]

    # Debye-Hueckel (dna2/debye.py:47-64; oxRNA2 uses the same term)
    if model in (2, 3):
        d = S["debye"]
        d_kt = d.get("kt")
        d_kt = kt if d_kt is None else d_kt
        d_salt = d.get("salt_conc")
        d_salt = _t(salt_conc) if d_salt is None else d_salt
        hce = d.get("half_charged_ends")
        hce = half_charged_ends if hce is None else bool(hce)
        lam = d["lambda_factor"] * xp.sqrt(d_kt / 0.1) / xp.sqrt(d_salt)
        r_high = 3.0 * lam
        pref = d["prefactor_coeff"] * d["q_eff"] ** 2
        v = pref * xp.exp(-r_high / lam) / r_high
        dv = -v * (1.0 / lam + 1.0 / r_high)
        bsm, r_cut = c1_match(r_high, v, dv)
        out.update(
            DH_KAPPA=1.0 / lam, DH_PREFACTOR=pref, DH_BSMOOTH=bsm, DH_RCUT=r_cut, DH_RHIGH=r_high,
            DH_HALF_CHARGED_ENDS=_t(1.0 if hce else 0.0),
        )
    else:
        out.update(
            DH_KAPPA=_t(1.0), DH_PREFACTOR=zero, DH_BSMOOTH=zero, DH_RCUT=zero, DH_RHIGH=zero,
            DH_HALF_CHARGED_ENDS=zero,
        )

    for i in range(4):
        for j in range(4):
            out[f"STCK_EPS_{i}{j}"] = eps_stack[i, j]
            out[f"HYDR_EPS_{i}{j}"] = eps_hb[i, j]
    tw = xp.ones(8) if term_weights is None else _t(term_weights)
    for k, name in enumerate(TERM_WEIGHT_NAMES):
        out[name] = tw[k]
    return out


NA1_SETS = ("dna", "rna", "drh")  # order of the three vectors of an oxNA system (model 4 of the C ABI)
NA1_UNBONDED_SECTIONS = ("unbonded_excluded_volume", "hydrogen_bonding", "cross_stacking", "coaxial_stacking", "debye")


def derive_flat_na1(sections_dna: dict, sections_rna: dict, sections_drh: dict, *, kt, salt_conc=0.5, half_charged_ends=False,
                    term_weights=None, numbers_ok: bool = False) -> dict[str, dict[str, torch.Tensor]]:
    """The three flat vectors of a hybrid DNA / RNA system (mythos/energy/na1/*.py): ``dna`` - DNA-DNA pairs, the oxDNA2
    sections in the oxDNA2 forms; ``rna`` - RNA-RNA pairs, oxRNA2; ``drh`` - DNA-RNA pairs: the five unbonded sections of
    mythos/input/na1/default_energy.toml in their oxDNA1 forms (cross-stacking with theta4, coaxial stacking with f5 of
    cos phi3 / phi4, Debye-Hueckel).  The hybrid kernels read the bonded entries and the geometry of neither: those
    places of the ``drh`` vector are filled from the oxRNA2 sections, which have the same form."""
    kw = dict(kt=kt, salt_conc=salt_conc, half_charged_ends=half_charged_ends, term_weights=term_weights, numbers_ok=numbers_ok)
    drh = {**{k: v for k, v in sections_rna.items() if k not in NA1_UNBONDED_SECTIONS}, **{k: sections_drh[k] for k in NA1_UNBONDED_SECTIONS}}
    return {"dna": derive_flat(2, sections_dna, **kw), "rna": derive_flat(3, sections_rna, **kw), "drh": derive_flat(3, drh, **kw)}


def pack_flat_na1(named3: dict, names: list[str]) -> torch.Tensor:
    return torch.cat([pack_flat(named3[k], names) for k in NA1_SETS])


def pack_flat(named: dict[str, torch.Tensor], names: list[str]) -> torch.Tensor:
    """Stack into the order the C ABI reports (mythos_oxdna_param_name)."""
    missing = [n for n in names if n not in named]
    if missing:
        raise KeyError(f"flat parameters not derived: {missing}")
    if not any(isinstance(named[n], torch.Tensor) for n in names):  # the numpy derivation (derive_flat, numbers_ok)
        return torch.from_numpy(np.array([float(named[n]) for n in names], dtype=np.float64))
    return torch.stack([_t(named[n]).reshape(()).to(F64) for n in names])


PI = math.pi
