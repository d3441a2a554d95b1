#!/bin/bash
# Round profile collection, run ON the GPU box from the repo root:
#   gpurun -- 'bash scripts/collect_profiles.sh r01'
# Writes the rocprofv3 kernel-trace summary of the default bench command, the PMC passes (FETCH_SIZE and
# WRITE_SIZE in separate passes, two SQ passes) and their per-kernel means under gpurun_out/<tag>/;
# copy gpurun_out/<tag>/profiles/* into profiles/ afterwards (tracked).
set -eo pipefail
tag=${1:-r01}
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out/profiles"
export TMPDIR=/tmp
short="--steps 500 --warmup 100 --cpu-steps 0 --no-second-dtype"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --cpu-steps 0 > $out/bench_profiled.log 2>&1
echo "[profiles] stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py $short > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py $short > $out/pmc_write.log 2>&1
echo "[profiles] traffic passes done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
  --output-format csv -d $out/pmc_sq1 -- python bench.py $short > $out/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR \
  --output-format csv -d $out/pmc_sq2 -- python bench.py $short > $out/pmc_sq2.log 2>&1
echo "[profiles] SQ passes done"
python scripts/collect_traffic.py $out/pmc_fetch $out/pmc_write --n 24000 --kernel 'md_step_kernel<float' --out $out/profiles/traffic.json
python - "$out" "$tag" <<'PY'
import glob, json, sys
import pandas as pd
out, tag = sys.argv[1], sys.argv[2]
res = {"kernel": "md_step_kernel<float, 2, false>", "command": "python bench.py --steps 500 --warmup 100 --cpu-steps 0 --no-second-dtype", "mean_per_dispatch": {}}
for d in ["pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"]:
    f = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)[0]
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains("md_step_kernel<float", regex=False)]
    for k, v in df.groupby("Counter_Name").Counter_Value.mean().items():
        res["mean_per_dispatch"][k] = v
m = res["mean_per_dispatch"]
res["derived"] = {
    "valu_insts_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
    "valu_busy_fraction_of_wave_cycles": m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"],
    "active_lanes_per_valu_inst": m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"]),
    "wait_any_fraction": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
    "lds_bank_conflict_fraction_of_lds_active": m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_ACTIVE_INST_LDS"], 1.0),
}
open(f"{out}/profiles/{tag}_md_step_pmc.json", "w").write(json.dumps(res, indent=1) + "\n")
ks = glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True)[0]
open(f"{out}/profiles/{tag}_bench_kernel_stats.csv", "w").write(open(ks).read())
line = [l for l in open(f"{out}/bench_profiled.log") if l.startswith("{")]
open(f"{out}/profiles/{tag}_bench_profiled_line.json", "w").write(line[-1] if line else "")
print(json.dumps(res["derived"]))
PY
echo "[profiles] written to $out/profiles"
