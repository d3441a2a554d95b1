#!/usr/bin/env python3
"""Fold one rocprofv3 kernel-trace run and one rocprofv3 --pmc run of the same command into a per-kernel summary:
dispatch count and mean duration (from the trace's begin / end stamps), and per dispatch the SQ counters and what
follows from them - VALU instructions per wavefront, active lanes per VALU instruction, share of wave cycles waiting,
and the VALU-issue bound of the launch (VALU wave-instructions / 1 024 SIMDs at 2 and at 4 cycles each).

    python scripts/profile_summary.py <trace_dir> <pmc_dir> <out.json> --command "..." [--min-share 0.02]
"""
import argparse
import glob
import json
import re

import pandas as pd

SIMDS, CLOCK_GHZ = 1024, 2.4


def short(name: str) -> str:
    name = re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", "").replace("mythos::", "")
    return re.sub(r"\(.*$", "", name)


ap = argparse.ArgumentParser()
ap.add_argument("trace_dir")
ap.add_argument("pmc_dir")
ap.add_argument("out")
ap.add_argument("--command", default="")
ap.add_argument("--min-share", type=float, default=0.02, help="kernels under this share of the traced GPU time are left out")
ap.add_argument("--alg-bytes", action="append", default=[], help="kernel-substring=bytes: algorithmic bytes per launch (HBM fraction)")
a = ap.parse_args()

tr = pd.read_csv(glob.glob(f"{a.trace_dir}/**/*kernel_trace.csv", recursive=True)[0])
tr["name"] = tr.Kernel_Name.map(short)
tr["us"] = (tr.End_Timestamp - tr.Start_Timestamp) / 1e3
total = tr.us.sum()
pm = pd.read_csv(glob.glob(f"{a.pmc_dir}/**/*counter_collection.csv", recursive=True)[0])
pm["name"] = pm.Kernel_Name.map(short)
alg = dict(x.rsplit("=", 1) for x in a.alg_bytes)
kernels = []
for name, g in sorted(tr.groupby("name"), key=lambda kv: -kv[1].us.sum()):
    if g.us.sum() < a.min_share * total:
        continue
    row = {"kernel": name, "dispatches": int(len(g)), "mean_us": float(g.us.mean()), "median_us": float(g.us.median()),
           "share_of_gpu_time": float(g.us.sum() / total), "rocprof_vgpr_count": int(g.VGPR_Count.iloc[0]) if "VGPR_Count" in g else None,
           "lds_bytes": int(g.LDS_Block_Size.iloc[0]) if "LDS_Block_Size" in g else None,
           "scratch_bytes": int(g.Scratch_Size.iloc[0]) if "Scratch_Size" in g else None}
    c = pm[pm.name == name]
    if len(c):
        m = c.groupby("Counter_Name").Counter_Value.mean()
        row["counters_mean_per_dispatch"] = {k: float(v) for k, v in m.items()}
        if {"SQ_INSTS_VALU", "SQ_WAVES"} <= set(m.index):
            row["valu_insts_per_wave"] = float(m.SQ_INSTS_VALU / m.SQ_WAVES)
            per_simd = m.SQ_INSTS_VALU / SIMDS
            row["valu_issue_bound_us"] = {"at_2_cycles": float(per_simd * 2 / (CLOCK_GHZ * 1e3)), "at_4_cycles": float(per_simd * 4 / (CLOCK_GHZ * 1e3))}
            row["valu_issue_fraction"] = {k: float(v / row["mean_us"]) for k, v in row["valu_issue_bound_us"].items()}
        if {"SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"} <= set(m.index):
            row["active_lanes_per_valu_inst"] = float(m.SQ_THREAD_CYCLES_VALU / (64.0 * m.SQ_ACTIVE_INST_VALU))
        if {"SQ_WAIT_ANY", "SQ_WAVE_CYCLES"} <= set(m.index):
            row["wait_any_fraction"] = float(m.SQ_WAIT_ANY / m.SQ_WAVE_CYCLES)
        if {"SQ_INSTS_SALU", "SQ_INSTS_VALU"} <= set(m.index):
            row["salu_per_valu"] = float(m.SQ_INSTS_SALU / m.SQ_INSTS_VALU)
    for sub, b in alg.items():
        if sub in name:
            row["algorithmic_bytes_per_launch"] = float(b)
            row["hbm_fraction_of_8TBs"] = float(b) / (row["mean_us"] * 1e-6) / 8e12
    kernels.append(row)
json.dump({"command": a.command, "gpu_time_traced_us": float(total), "kernels": kernels}, open(a.out, "w"), indent=1)
for r in kernels:
    print(f"{r['kernel'][:70]:70s} n={r['dispatches']:6d} mean {r['mean_us']:9.2f} us  valu/wave {r.get('valu_insts_per_wave', float('nan')):8.1f} "
          f"lanes {r.get('active_lanes_per_valu_inst', float('nan')):.2f} wait {r.get('wait_any_fraction', float('nan')):.2f}")
