"""Pin the MARTINI oracle to the GROMACS goldens the reference's tests use
(mythos/energy/martini/m2/tests/test_{lj,bond,angle}.py, allclose rtol 1e-5)."""

import numpy as np
import torch

from oracle import martini_oracle as mo
from tests import martini_helpers as MH


def test_topology_from_text_files():
    s = MH.system()
    top = s["top"]
    assert len(top.atom_types) == 1280 and top.bonded_neighbors.shape == (128 * 9, 2) and top.angles.shape == (128 * 6, 3)
    assert top.bond_names[0] == "DMPC_NC3_PO4" and top.angle_names[0] == "DMPC_PO4_GL1_GL2"
    x, box, e = MH.frames("lj")
    assert x.shape == (10, 1280, 3) and box.shape == (10, 3) and e.shape == (10,)


def test_lj_matches_gromacs():
    s = MH.system()
    x, box, e = MH.frames("lj")
    sig, eps = torch.as_tensor(s["sigma"]), torch.as_tensor(s["eps"])
    got = [float(mo.lj_energy(torch.as_tensor(x[f]), torch.as_tensor(box[f]), s["types"], sig, eps, s["top"].bonded_neighbors)) for f in range(10)]
    np.testing.assert_allclose(got, e, rtol=1e-5, atol=1e-8)


def test_bond_matches_gromacs():
    s = MH.system()
    x, box, e = MH.frames("bond")
    got = [float(mo.bond_energy(torch.as_tensor(x[f]), torch.as_tensor(box[f]), s["top"].bonded_neighbors, torch.as_tensor(s["bond_k"]), torch.as_tensor(s["bond_r0"]))) for f in range(10)]
    np.testing.assert_allclose(got, e, rtol=1e-5, atol=1e-8)


def test_g96_angle_matches_gromacs():
    s = MH.system()
    x, box, e = MH.frames("angle")
    got = [float(mo.angle_energy(torch.as_tensor(x[f]), torch.as_tensor(box[f]), s["top"].angles, torch.as_tensor(s["angle_k"]), torch.as_tensor(s["angle_t0"]), True)) for f in range(10)]
    np.testing.assert_allclose(got, e, rtol=1e-5, atol=1e-8)
