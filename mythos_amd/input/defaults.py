"""Default oxDNA1 / oxDNA2 / oxRNA2 parameter tables, restated as Python data.

Values restate the reference's TOML tables
(mythos/input/dna1/default_energy.toml, mythos/input/dna2/default_energy.toml, mythos/input/rna2/default_energy.toml,
mythos/input/dna{1,2}/default_simulation.toml); expressions the TOML writes as
strings ("pi - 2.35") are evaluated here with ``math.pi`` in fp64.  A TOML file
with the same section/key layout can be loaded with :func:`parse_toml`.
"""

from __future__ import annotations

import ast
import copy
import math
import operator
from pathlib import Path

PI = math.pi

_COMMON_EXC_BONDED = {
    "eps_exc": 2.0,
    "dr_star_base": 0.32,
    "sigma_base": 0.33,
    "sigma_back_base": 0.515,
    "sigma_base_back": 0.515,
    "dr_star_back_base": 0.50,
    "dr_star_base_back": 0.50,
}

_COMMON_EXC_UNBONDED = {
    "eps_exc": 2.0,
    "dr_star_base": 0.32,
    "sigma_base": 0.33,
    "dr_star_back_base": 0.50,
    "sigma_back_base": 0.515,
    "dr_star_base_back": 0.50,
    "sigma_base_back": 0.515,
    "dr_star_backbone": 0.675,
    "sigma_backbone": 0.70,
}


def _stacking(eps_base: float, kt_coeff: float) -> dict:
    return {
        "eps_stack_base": eps_base,
        "eps_stack_kt_coeff": kt_coeff,
        "a_stack": 6.0,
        "dr0_stack": 0.4,
        "dr_c_stack": 0.9,
        "dr_low_stack": 0.32,
        "dr_high_stack": 0.75,
        "a_stack_4": 1.30,
        "theta0_stack_4": 0.0,
        "delta_theta_star_stack_4": 0.8,
        "a_stack_5": 0.90,
        "theta0_stack_5": 0.0,
        "delta_theta_star_stack_5": 0.95,
        "a_stack_6": 0.90,
        "theta0_stack_6": 0.0,
        "delta_theta_star_stack_6": 0.95,
        "a_stack_1": 2.00,
        "neg_cos_phi1_star_stack": -0.65,
        "a_stack_2": 2.00,
        "neg_cos_phi2_star_stack": -0.65,
    }


def _hydrogen_bonding(eps_hb: float) -> dict:
    d = {
        "dr_low_hb": 0.34,
        "dr_high_hb": 0.70,
        "eps_hb": eps_hb,
        "a_hb": 8.0,
        "dr0_hb": 0.4,
        "dr_c_hb": 0.75,
    }
    for k, (t0, ds, a) in {
        1: (0.0, 0.70, 1.50),
        2: (0.0, 0.70, 1.50),
        3: (0.0, 0.70, 1.50),
        4: (PI, 0.70, 0.46),
        7: (PI / 2, 0.45, 4.0),
        8: (PI / 2, 0.45, 4.0),
    }.items():
        d[f"theta0_hb_{k}"] = t0
        d[f"delta_theta_star_hb_{k}"] = ds
        d[f"a_hb_{k}"] = a
    return d


def _cross_stacking() -> dict:
    d = {
        "dr_low_cross": 0.495,
        "dr_high_cross": 0.655,
        "k_cross": 47.5,
        "r0_cross": 0.575,
        "dr_c_cross": 0.675,
    }
    for k, (t0, ds, a) in {
        1: (PI - 2.35, 0.58, 2.25),
        2: (1.00, 0.68, 1.70),
        3: (1.00, 0.68, 1.70),
        4: (0.0, 0.65, 1.50),
        7: (0.875, 0.68, 1.70),
        8: (0.875, 0.68, 1.70),
    }.items():
        d[f"theta0_cross_{k}"] = t0
        d[f"delta_theta_star_cross_{k}"] = ds
        d[f"a_cross_{k}"] = a
    return d


def _coaxial(k_coax: float, theta0_1: float) -> dict:
    d = {
        "dr_low_coax": 0.22,
        "dr_high_coax": 0.58,
        "k_coax": k_coax,
        "dr0_coax": 0.4,
        "dr_c_coax": 0.6,
    }
    for k, (t0, ds, a) in {
        4: (0.0, 0.8, 1.30),
        1: (theta0_1, 0.65, 2.00),
        5: (0.0, 0.95, 0.90),
        6: (0.0, 0.95, 0.90),
    }.items():
        d[f"theta0_coax_{k}"] = t0
        d[f"delta_theta_star_coax_{k}"] = ds
        d[f"a_coax_{k}"] = a
    return d


_FENE = {"eps_backbone": 2.0, "delta_backbone": 0.25, "fmax": 500.0, "finf": 4.0}

DNA1_ENERGY = {
    "geometry": {"com_to_stacking": 0.34, "com_to_hb": 0.4, "com_to_backbone": -0.4},
    "fene": {**_FENE, "r0_backbone": 0.7525},
    "bonded_excluded_volume": dict(_COMMON_EXC_BONDED),
    "stacking": _stacking(1.3448, 2.6568),
    "unbonded_excluded_volume": dict(_COMMON_EXC_UNBONDED),
    "hydrogen_bonding": _hydrogen_bonding(1.077),
    "cross_stacking": _cross_stacking(),
    "coaxial_stacking": {
        **_coaxial(46.0, PI - 0.60),
        "cos_phi3_star_coax": -0.65,
        "a_coax_3p": 2.0,
        "cos_phi4_star_coax": -0.65,
        "a_coax_4p": 2.0,
    },
}

DNA2_ENERGY = {
    "geometry": {
        "com_to_stacking": 0.34,
        "com_to_hb": 0.4,
        "com_to_backbone_x": -0.34,
        "com_to_backbone_y": 0.3408,
        "com_to_backbone_dna1": -0.4,
    },
    "fene": {**_FENE, "r0_backbone": 0.7564},
    "bonded_excluded_volume": dict(_COMMON_EXC_BONDED),
    "stacking": _stacking(1.3523, 2.6717),
    "unbonded_excluded_volume": dict(_COMMON_EXC_UNBONDED),
    "hydrogen_bonding": _hydrogen_bonding(1.0678),
    "cross_stacking": _cross_stacking(),
    "coaxial_stacking": {**_coaxial(58.5, PI - 0.25), "a_coax_1_f6": 40.0, "b_coax_1_f6": PI - 0.025},
    "debye": {"q_eff": 0.815, "lambda_factor": 0.3616455075438555, "prefactor_coeff": 0.08173808693529228},
}

def _rna2_stacking() -> dict:
    d = {"eps_stack_base": 1.40206, "eps_stack_kt_coeff": 2.77, "a_stack": 6.0, "dr0_stack": 0.43, "dr_c_stack": 0.93,
         "dr_low_stack": 0.35, "dr_high_stack": 0.78, "a_stack_1": 2.00, "neg_cos_phi1_star_stack": -0.65, "a_stack_2": 2.00,
         "neg_cos_phi2_star_stack": -0.65}
    for k, (t0, ds, a) in {5: (0.0, 0.95, 0.90), 6: (0.0, 0.95, 0.90), 9: (0.0, 0.8, 1.3), 10: (0.0, 0.8, 1.3)}.items():
        d[f"theta0_stack_{k}"], d[f"delta_theta_star_stack_{k}"], d[f"a_stack_{k}"] = t0, ds, a
    return d


def _rna2_cross_stacking() -> dict:
    d = {"k_cross": 59.9626, "r0_cross": 0.5, "dr_c_cross": 0.6, "dr_low_cross": 0.42, "dr_high_cross": 0.58}
    for k, (t0, ds, a) in {1: (0.505, 0.58, 2.25), 2: (1.266, 0.68, 1.70), 3: (1.266, 0.68, 1.70), 7: (0.309, 0.68, 1.70),
                           8: (0.309, 0.68, 1.70)}.items():
        d[f"theta0_cross_{k}"], d[f"delta_theta_star_cross_{k}"], d[f"a_cross_{k}"] = t0, ds, a
    return d


def _rna2_coaxial() -> dict:
    d = {"k_coax": 80.0, "dr0_coax": 0.5, "dr_c_coax": 0.6, "dr_low_coax": 0.42, "dr_high_coax": 0.58,
         "a_coax_3p": 2.0, "cos_phi3_star_coax": -0.65, "a_coax_4p": 2.0, "cos_phi4_star_coax": -0.65}
    for k, (t0, ds, a) in {4: (0.151, 0.8, 1.30), 1: (2.592, 0.65, 2.00), 5: (0.685, 0.95, 0.90), 6: (0.685, 0.95, 0.90)}.items():
        d[f"theta0_coax_{k}"], d[f"delta_theta_star_coax_{k}"], d[f"a_coax_{k}"] = t0, ds, a
    return d


# oxRNA2 (mythos/input/rna2/default_energy.toml): the oxDNA1 term set with its own numbers, a backbone site off a1 and
# a3, separate 3' / 5' stacking sites, the p3 / p5 vectors of its stacking term, and Debye-Hueckel
RNA2_ENERGY = {
    "geometry": {
        "pos_stack": 0.34, "pos_base": 0.4, "pos_back_a1": -0.4, "pos_back_a3": 0.2, "pos_back_a2": 0.0,
        "pos_stack_3_a1": 0.4, "pos_stack_3_a2": 0.1, "pos_stack_5_a1": 0.124906078525, "pos_stack_5_a2": -0.00866274917473,
        "p5_x": -0.104402, "p5_y": -0.841783, "p5_z": 0.529624, "p3_x": -0.462510, "p3_y": -0.528218, "p3_z": 0.712089,
    },
    "fene": {**_FENE, "r0_backbone": 0.761070781051},
    "bonded_excluded_volume": dict(_COMMON_EXC_BONDED),
    "stacking": _rna2_stacking(),
    "unbonded_excluded_volume": dict(_COMMON_EXC_UNBONDED),
    "hydrogen_bonding": _hydrogen_bonding(0.870439),
    "cross_stacking": _rna2_cross_stacking(),
    "coaxial_stacking": _rna2_coaxial(),
    "debye": {"q_eff": 1.26, "lambda_factor": 0.3667258, "prefactor_coeff": 0.05404383975812547},
}

# oxNA (mythos/input/na1/default_energy.toml): the numbers of a DNA-RNA hybrid pair, unbonded terms only, every term in its
# oxDNA1 functional form; DNA-DNA pairs take DNA2_ENERGY and RNA-RNA pairs RNA2_ENERGY (na1/tests/test_integration.py:104-141)
NA1_DRH_ENERGY = {
    "unbonded_excluded_volume": dict(_COMMON_EXC_UNBONDED),
    "hydrogen_bonding": _hydrogen_bonding(1.5),
    "cross_stacking": {**_cross_stacking(), "k_cross": 44.535},
    "coaxial_stacking": {**_coaxial(46.0, PI - 0.60), "cos_phi3_star_coax": -0.65, "a_coax_3p": 2.0, "cos_phi4_star_coax": -0.65,
                         "a_coax_4p": 2.0},
    "debye": {"q_eff": 1.26, "lambda_factor": 0.3667258, "prefactor_coeff": 0.05404383975812547},
}

_SIM_COMMON = {
    "kT": 296.15 * 0.1 / 300.0,
    "dt": 5e-3,
    "diff_coef": 2.5,
    "rot_diff_coef": 7.5,
    "nucleotide_mass": 1.0,
    "moment_of_inertia": [1.0, 1.0, 1.0],
    "n_steps": 100,
    "checkpoint_interval": 0,
}
DNA1_SIMULATION = dict(_SIM_COMMON)
DNA2_SIMULATION = {**_SIM_COMMON, "salt_conc": 0.5, "half_charged_ends": 1}
# the reference ships no rna2 simulation TOML; its golden run (data/test-data/rna2/*/input) is at salt 1.0, whole end charges
RNA2_SIMULATION = {**_SIM_COMMON, "salt_conc": 1.0, "half_charged_ends": 0}


def default_configs_for(base: str) -> tuple[dict, dict]:
    """(simulation config, energy config) deep copies; ``base`` is "dna1", "dna2", "rna2" or "na1" (three sets).

    Reference: mythos/energy/utils.py:135-148.
    """
    if base == "dna1":
        return copy.deepcopy(DNA1_SIMULATION), copy.deepcopy(DNA1_ENERGY)
    if base == "dna2":
        return copy.deepcopy(DNA2_SIMULATION), copy.deepcopy(DNA2_ENERGY)
    if base == "rna2":
        return copy.deepcopy(RNA2_SIMULATION), copy.deepcopy(RNA2_ENERGY)
    if base == "na1":  # three parameter sets: DNA-DNA, RNA-RNA and hybrid pairs (the goldens: salt 0.5, whole end charges)
        return ({**_SIM_COMMON, "salt_conc": 0.5, "half_charged_ends": 0},
                {"dna": copy.deepcopy(DNA2_ENERGY), "rna": copy.deepcopy(RNA2_ENERGY), "drh": copy.deepcopy(NA1_DRH_ENERGY)})
    raise ValueError(f"unknown model '{base}' (expected 'dna1', 'dna2', 'rna2' or 'na1')")


def energy_section_names(base: str) -> tuple:
    """Section names of the default energy configuration of "dna1" / "dna2" / "rna2" (no copy made)."""
    return tuple({"dna1": DNA1_ENERGY, "dna2": DNA2_ENERGY, "rna2": RNA2_ENERGY}[base])


# ---------------------------------------------------------------------------------------------
# TOML with arithmetic strings (reference: mythos/input/toml.py:21-57, which uses sympy)
# ---------------------------------------------------------------------------------------------

_BINOPS = {
    ast.Add: operator.add,
    ast.Sub: operator.sub,
    ast.Mult: operator.mul,
    ast.Div: operator.truediv,
    ast.Pow: operator.pow,
}
_NAMES = {"pi": math.pi, "e": math.e}
_FUNCS = {"sqrt": math.sqrt, "cos": math.cos, "sin": math.sin, "exp": math.exp, "log": math.log}


def _eval_node(node):
    if isinstance(node, ast.Expression):
        return _eval_node(node.body)
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
        return float(node.value)
    if isinstance(node, ast.Name) and node.id in _NAMES:
        return _NAMES[node.id]
    if isinstance(node, ast.BinOp) and type(node.op) in _BINOPS:
        return _BINOPS[type(node.op)](_eval_node(node.left), _eval_node(node.right))
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _eval_node(node.operand)
        return -v if isinstance(node.op, ast.USub) else v
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in _FUNCS:
        return _FUNCS[node.func.id](*[_eval_node(a) for a in node.args])
    raise ValueError("unsupported expression")


def parse_str(value: str):
    """A float if ``value`` is a number or an arithmetic expression in pi, else the string."""
    try:
        return float(value)
    except ValueError:
        try:
            return float(_eval_node(ast.parse(value.replace("^", "**"), mode="eval")))
        except (ValueError, SyntaxError, TypeError, ZeroDivisionError):
            return value


def _parse_value(value):
    if isinstance(value, str):
        return parse_str(value)
    if isinstance(value, list):
        return [_parse_value(v) for v in value]
    if isinstance(value, dict):
        return {k: _parse_value(v) for k, v in value.items()}
    return value


def parse_toml(file_path, key: str | None = None) -> dict:
    """Parse a TOML parameter file; string leaves are evaluated as arithmetic."""
    try:
        import tomllib as toml
    except ImportError:  # Python 3.10
        import tomli as toml
    with Path(file_path).open("rb") as f:
        cfg = toml.load(f)
    if key is not None:
        if key not in cfg:
            raise ValueError(f"Missing entry {key} in TOML file")
        cfg = cfg[key]
    return _parse_value(cfg)
