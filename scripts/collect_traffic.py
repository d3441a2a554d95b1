#!/usr/bin/env python3
"""Fold the rocprofv3 PMC passes of ``bench.py`` into profiles/traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
    python scripts/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --n 24000 --dtype f32

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled
(/opt/skills/guides/MI355X_MICROARCH.md, HBM).  The result is the mean over the md_step_kernel dispatches.
"""
import argparse, glob, json
from pathlib import Path

import pandas as pd


def mean_counter(d, name, kernel):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains(kernel, regex=False) & (df.Counter_Name == name)]
    return float(df.Counter_Value.mean()), int(len(df))


ap = argparse.ArgumentParser()
ap.add_argument("fetch_dir")
ap.add_argument("write_dir")
ap.add_argument("--n", type=int, required=True)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--kernel", default="md_step_kernel")
ap.add_argument("--out", default=str(Path(__file__).resolve().parent.parent / "profiles" / "traffic.json"))
a = ap.parse_args()
fetch_kib, nf = mean_counter(a.fetch_dir, "FETCH_SIZE", a.kernel)
write_kib, nw = mean_counter(a.write_dir, "WRITE_SIZE", a.kernel)
out = {
    "kernel": a.kernel, "n_nucleotides": a.n, "dtype": a.dtype,
    "FETCH_SIZE_KiB_mean": fetch_kib, "WRITE_SIZE_KiB_mean": write_kib, "dispatches": [nf, nw],
    "correction": "2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE",
    "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
}
Path(a.out).write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out))
