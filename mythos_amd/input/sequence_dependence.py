"""oxDNA sequence-dependence files -> the weight matrices of the stacking and hydrogen-bonding terms.

Mirror of the reference's ``mythos/input/sequence_dependence.py:12-51`` (same name, same keys, same failure on a
missing entry): a file of ``KEY = VALUE`` lines - white space ignored, values may carry oxDNA's ``f`` suffix - with

* ``STCK_X_Y`` for the sixteen (5' base, 3' base) stacking strengths and ``STCK_FACT_EPS``,
* ``HYDR_A_T`` or ``HYDR_T_A``, ``HYDR_G_C`` or ``HYDR_C_G`` (oxDNA sets both members of a pair to one value).

The result goes into ``with_params`` / the configurations as it does in the reference::

    w = read_ss_weights("oxDNA2_sequence_dependent_parameters.txt")
    energy_fn = energy_fn.with_params(ss_stack_weights=w["ss_stack_weights"], eps_stack_kt_coeff=w["eps_stack_kt_coeff"],
                                      ss_hb_weights=w["ss_hb_weights"])
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

DNA_ALPHA = "ACGT"  # the reference's base order (mythos/utils/constants.py:5-11)
N_IDX = {b: k for k, b in enumerate(DNA_ALPHA)}


def read_ss_weights(file) -> dict[str, torch.Tensor]:
    """-> {"eps_stack_kt_coeff": scalar, "ss_stack_weights": (4, 4), "ss_hb_weights": (4, 4)}, float64 tensors.

    A stacking entry that is missing, or a file with neither member of a Watson-Crick pair, raises ``KeyError`` naming it
    (the reference's behaviour); entries the terms do not use (``HYDR_G_T`` of the RNA files, ...) are ignored."""
    param_map: dict[str, float] = {}
    with Path(file).open("r") as f:
        for line in f:
            kv = line.strip().replace(" ", "")
            if kv:
                key, val = kv.split("=")
                param_map[key] = float(val.replace("f", ""))
    stack = np.zeros((4, 4), dtype=np.float64)
    for i, row in enumerate(DNA_ALPHA):
        for j, col in enumerate(DNA_ALPHA):
            stack[i, j] = param_map[f"STCK_{row}_{col}"]
    hb_a_t = param_map["HYDR_A_T"] if "HYDR_A_T" in param_map else param_map["HYDR_T_A"]
    hb_g_c = param_map["HYDR_G_C"] if "HYDR_G_C" in param_map else param_map["HYDR_C_G"]
    hb = np.zeros((4, 4), dtype=np.float64)
    hb[N_IDX["A"], N_IDX["T"]] = hb[N_IDX["T"], N_IDX["A"]] = hb_a_t
    hb[N_IDX["G"], N_IDX["C"]] = hb[N_IDX["C"], N_IDX["G"]] = hb_g_c
    return {
        "eps_stack_kt_coeff": torch.tensor(param_map["STCK_FACT_EPS"], dtype=torch.float64),
        "ss_stack_weights": torch.as_tensor(stack),
        "ss_hb_weights": torch.as_tensor(hb),
    }
