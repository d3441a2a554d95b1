"""Provenance of the golden fixtures: the DATA files the reference ships for its own tests, copied unmodified.

Run in the build container (where /root/reference exists) to re-create ``tests/golden/``:

    python tests/golden/copy_fixtures.py            # copies and verifies byte-for-byte
    python tests/golden/copy_fixtures.py --check    # only verifies

oxDNA goldens (``data/test-data/dna{1,2}/<case>/``): topology, the 100-frame trajectory, oxDNA's own total and
split energies, per-pair energies and the oxDNA input file they were produced with (reference tests:
mythos/energy/dna1/tests/test_integration.py, mythos/energy/dna2/tests/test_integration.py).
MARTINI goldens (``data/test-data/martini/energy/m2/``): GROMACS trajectory, per-term energies, parameter JSON
(reference tests: mythos/energy/martini/m2/tests/test_{lj,bond,angle}.py); the system definition comes from the
text files next to the reference's binary ``.tpr`` (``data/templates/martini/m2/DMPC/273K/``).
``oxdna_model_constants.json`` is produced by ``make_model_constants.py``.  No reference source code is copied.
"""

import argparse
import filecmp
import shutil
from pathlib import Path

REF = Path("/root/reference/data")
DST = Path(__file__).resolve().parent

OXDNA_FILES = ("generated.top", "output.dat", "energy.dat", "split_energy.dat", "pair.dat", "input")
CASES = {
    "dna1": ("simple-helix", "simple-coax", "simple-helix-seq-dep", "helix-4bp"),
    "dna2": ("simple-helix", "simple-coax", "simple-helix-half-charged-ends"),
    # oxRNA2 (reference tests: mythos/energy/rna2/tests/test_integration.py)
    "rna2": ("simple-helix-12bp", "simple-coax"),
    # oxNA hybrid DNA / RNA (reference tests: mythos/energy/na1/tests/test_integration.py); without pair.dat (0.4 MB each)
    "na1": ("simple-helix-dna-dna", "simple-helix-rna-rna", "simple-helix-dna-rna", "simple-helix-rna-dna",
            "simple-coax-dna-dna-dna", "simple-coax-rna-rna-rna", "simple-coax-dna-dna-rna"),
}
# System definitions the reference ships for its examples (examples/jaxmd/simulation.ipynb, the tutorials): relaxed, i.e.
# thermally distorted, conformations - a 500 bp duplex (1 000 nt, box 350) and the 110 bp WLC system (220 nt).  Inputs only:
# the expected values of the tests on them come from the oracle.
SYSDEFS = {
    "sys-defs/persistence-length-500bp": ("sys-defs/persistence-length-500bp", ("init.top", "relaxed.dat")),
    "sys-defs/wlc-fit": ("sys-defs/wlc-fit", ("generated.top", "generated.dat")),
    # ... and the small ones of the melting-temperature examples: hairpins (ONE strand whose ends pair: the stem's base
    # pairs are unbonded pairs inside a strand, the loop is unpaired), the 8 bp duplex bound, relaxed and with its strands apart
    "sys-defs/hairpins/4bp_stem_8nt_loop": ("sys-defs/hairpins/4bp_stem_8nt_loop", ("sys.top", "init.conf")),
    "sys-defs/hairpins/6bp_stem_6nt_loop": ("sys-defs/hairpins/6bp_stem_6nt_loop", ("sys.top", "init_bound.conf", "init_unbound.conf")),
    "sys-defs/simple-helix": ("sys-defs/simple-helix", ("sys.top", "bound.conf", "bound_relaxed.conf", "unbound.conf")),
}
# oxDNA regression runs the reference ships without a test of its own (data/test-data/regr-*): oxDNA2, half-charged ends,
# average sequence.  regr-circle: ONE circular 50-nt strand, 40 frames, oxDNA's split energies.  regr-burns-natnano-2015: a
# membrane channel of six circular 50-nt strands (nanobase.org/structure/13; notes.txt), the first TRIM_FRAMES of its 40 frames
# (3.3 MB otherwise) - a trimmed file is checked to be a prefix of the reference's.
REGR = {
    "test-data/regr-circle": ("regr/circle", ("sys.top", "output.dat", "split_energy.dat", "energy.dat", "input")),
    "test-data/regr-burns-natnano-2015": ("regr/burns-natnano-2015", ("sys.top", "output.dat", "split_energy.dat", "energy.dat", "input", "notes.txt")),
}
# ... and three more oxDNA2 runs with oxDNA's split energies that no reference test reads: sequence-dependent stacking / H-bond
# weights from oxDNA2's parameter file, a three-strand coaxial stack with the strands in the other order, a 12 bp duplex.
REGR.update({
    "test-data/simple-helix-oxdna2-ss": ("regr/simple-helix-oxdna2-ss", ("generated.top", "output.dat", "split_energy.dat", "energy.dat", "input",
                                                                           "oxDNA2_sequence_dependent_parameters.txt")),
    "test-data/simple-coax-oxdna2-rev": ("regr/simple-coax-oxdna2-rev", ("generated.top", "output.dat", "split_energy.dat", "energy.dat", "input")),
    "test-data/simple-helix-oxdna2-12bp": ("regr/simple-helix-oxdna2-12bp", ("sys.top", "output.dat", "split_energy.dat", "energy.dat", "input")),
    # oxRNA2 with HALF-charged strand ends (salt 1.0): the reference's rna2 tests run on whole end charges only
    "test-data/simple-helix-rna2-12bp-half-charged-ends": ("regr/simple-helix-rna2-12bp-half-charged-ends",
                                                             ("sys.top", "output.dat", "split_energy.dat", "energy.dat", "input")),
})
# LAMMPS (Sep 2021, oxdna2 pair styles) runs of a 40 bp duplex under tension and torque at T = 0.1, salt 0.15 - an
# implementation independent of oxDNA's, at a temperature and a salt concentration no other golden has: the TacoxDNA conversion
# of its dump (data.oxdna, first TRIM_FRAMES frames), its log with the per-term energies of every dumped step, its input.
# "-sa": average-sequence strengths, the other: LAMMPS's sequence-dependent tables.
REGR.update({
    "test-data/lammps-oxdna2-40bp-sa": ("regr/lammps-oxdna2-40bp-sa", ("data.top", "data.oxdna", "log.lammps", "in")),
    "test-data/lammps-oxdna2-40bp": ("regr/lammps-oxdna2-40bp", ("data.top", "data.oxdna", "log.lammps", "in", "notes.txt")),
})
# oxRNA2's model constants as oxDNA's external-model file lists them, the dependent smoothing constants included (the oxRNA2
# counterpart of data/templates/model_template.h, which holds the DNA ones)
REGR["test-data/regr-rna2-2ht-293.15-sa"] = ("regr/rna2-external-model", ("external_model.txt",))
# 60 bp duplexes (120 nt) with oxDNA's total potential energy per printed step: oxDNA1 (only the last configuration was kept) and
# oxDNA2 with half-charged ends (first TRIM_FRAMES configurations)
REGR["test-data/simple-helix-60bp"] = ("regr/simple-helix-60bp", ("sys.top", "last_conf.dat", "energy.dat", "input"))
REGR["test-data/simple-helix-60bp-oxdna2"] = ("regr/simple-helix-60bp-oxdna2", ("sys.top", "output.dat", "energy.dat", "input"))
# the sequence-dependence files the reference ships for its users (input/sequence_dependence.py reads them)
REGR["seq-specific"] = ("seq-specific", ("seq_oxdna1.txt", "seq_oxdna2.txt"))
TRIM_FRAMES = {("regr/burns-natnano-2015", "output.dat"): 10, ("regr/simple-helix-60bp-oxdna2", "output.dat"): 8, ("regr/lammps-oxdna2-40bp-sa", "data.oxdna"): 40,
               ("regr/lammps-oxdna2-40bp", "data.oxdna"): 12, ("regr/simple-helix-oxdna2-ss", "output.dat"): 25,
               ("regr/simple-coax-oxdna2-rev", "output.dat"): 25, ("regr/simple-helix-oxdna2-12bp", "output.dat"): 25,
               ("regr/simple-helix-rna2-12bp-half-charged-ends", "output.dat"): 25}
SKIP = {"na1": ("pair.dat",)}
EXTRA = {("dna1", "simple-helix-seq-dep"): ("generated-new.top", "seq_dep.dat"),
         # 8-nt duplex of the probabilistic-sequence tests (mythos/energy/dna1/tests/test_expected_energies.py:162-328)
         ("dna1", "helix-4bp"): ("sys.top",)}
MARTINI = {
    "test-data/martini/energy/m2/lj": ("martini/m2/lj", ("test.trr", "lj.xvg", "ljconf.json")),
    "test-data/martini/energy/m2/bond": ("martini/m2/bond", ("bond.xvg", "bond_params.json")),
    "test-data/martini/energy/m2/angle": ("martini/m2/angle", ("test.trr", "angle.xvg", "angle_params.json")),
    "templates/martini/m2/DMPC/273K": ("martini/template", ("membrane.gro", "topol.top")),
    # MARTINI-3 harmonic angles: DOPC bilayer + water, topology only as a GROMACS run input file
    # (reference test: mythos/energy/martini/m3/tests/test_angle_m3.py:61-77)
    "test-data/martini/energy/m3/angle": ("martini/m3/angle", ("test.trr", "test.tpr", "angle.xvg", "angle_params_rad.json")),
}


def pairs():
    for model, cases in CASES.items():
        for case in cases:
            for name in (*OXDNA_FILES, *EXTRA.get((model, case), ())):
                if name in SKIP.get(model, ()):
                    continue
                src = REF / "test-data" / model / case / name
                if src.exists():
                    yield src, DST / model / case / name
    for src_dir, (dst_dir, names) in (*MARTINI.items(), *SYSDEFS.items(), *REGR.items()):
        for name in names:
            yield REF / src_dir / name, DST / dst_dir / name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    bad = 0
    for src, dst in pairs():
        frames = TRIM_FRAMES.get((str(dst.parent.relative_to(DST)), dst.name))
        if frames is not None:  # the first `frames` frames of an oxDNA trajectory ("t = ..." opens a frame)
            text = src.read_text()
            starts = [m for m in range(len(text)) if text.startswith("t = ", m) and (m == 0 or text[m - 1] == "\n")]
            head = text if len(starts) <= frames else text[: starts[frames]]
            if not args.check:
                dst.parent.mkdir(parents=True, exist_ok=True)
                dst.write_text(head)
            if not dst.exists() or dst.read_text() != head:
                print("MISMATCH (trimmed)", dst.relative_to(DST))
                bad += 1
            continue
        if not args.check:
            dst.parent.mkdir(parents=True, exist_ok=True)
            shutil.copyfile(src, dst)
        if not dst.exists() or not filecmp.cmp(src, dst, shallow=False):
            print("MISMATCH", dst.relative_to(DST))
            bad += 1
    print("ok" if bad == 0 else f"{bad} file(s) differ")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
