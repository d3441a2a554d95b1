"""Loading of the MARTINI-2 GROMACS fixtures (tests/golden/martini) for oracle and GPU tests."""

from __future__ import annotations

import functools
import json

import numpy as np

from mythos_amd.input import gromacs
from tests.helpers import GOLDEN

MG = GOLDEN / "martini"


@functools.lru_cache(maxsize=None)
def system():
    top = gromacs.MartiniTopology.from_top(MG / "template" / "topol.top")
    lj = json.loads((MG / "m2" / "lj" / "ljconf.json").read_text())
    bead_types = sorted({t for k in lj for t in k.split("_")[2:4]})
    idx = {t: i for i, t in enumerate(bead_types)}

    def table(prefix):
        m = np.zeros((len(bead_types), len(bead_types)))
        for a in bead_types:
            for b in bead_types:
                v = lj.get(f"lj_{prefix}_{a}_{b}", lj.get(f"lj_{prefix}_{b}_{a}"))
                m[idx[a], idx[b]] = v
        return m

    types = np.array([idx[t] for t in top.atom_types], dtype=np.int32)
    bp = json.loads((MG / "m2" / "bond" / "bond_params.json").read_text())
    ap = json.loads((MG / "m2" / "angle" / "angle_params.json").read_text())
    bond_k = np.array([bp["bond_k_" + n] for n in top.bond_names])
    bond_r0 = np.array([bp["bond_r0_" + n] for n in top.bond_names])
    angle_k = np.array([ap["angle_k_" + n] for n in top.angle_names])
    angle_t0 = np.deg2rad(np.array([ap["angle_theta0_" + n] for n in top.angle_names]))
    return dict(top=top, types=types, sigma=table("sigma"), eps=table("epsilon"), bond_k=bond_k, bond_r0=bond_r0,
                angle_k=angle_k, angle_t0=angle_t0, bead_types=bead_types, lj_params=lj, bond_params=bp, angle_params=ap)


@functools.lru_cache(maxsize=None)
def frames(which: str):
    """(positions (10,1280,3), box (10,3), golden energies (10,)) for 'lj' | 'bond' | 'angle'."""
    trr = MG / "m2" / ("angle" if which == "angle" else "lj") / "test.trr"
    x, box, _ = gromacs.read_trr(trr)
    e = gromacs.read_xvg(MG / "m2" / which / f"{which}.xvg")[1:]
    return x, box, e
