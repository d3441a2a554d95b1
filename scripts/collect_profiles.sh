#!/bin/bash
# Round profile collection, run ON the GPU box from the repo root:   gpurun -- 'bash scripts/collect_profiles.sh r04'
# For every kernel that ships a number in DESIGN section 7: a rocprofv3 kernel trace of the command that measures it and
# ONE PMC pass of the same command (SQ counters), folded into profiles/<tag>_<case>.json by scripts/profile_summary.py;
# for the headline kernel also the FETCH_SIZE / WRITE_SIZE passes (profiles/traffic.json) and the trace's own
# --stats table.  The program after `--` is python itself.  Copy gpurun_out/<tag>/profiles/* into profiles/ afterwards.
set -eo pipefail
tag=${1:-r04}
part=${2:-all}   # "md", "rest" or "all": the whole collection is ~20 profiled runs, two gpurun calls fit it comfortably
out=gpurun_out/$tag
[ "$part" = rest ] || rm -rf "$out"
mkdir -p "$out/profiles"
export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"
one() {  # name, "alg-bytes args", command...
  local name=$1 alg=$2; shift 2
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${name}_trace -- "$@" > $out/${name}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $out/${name}_pmc -- "$@" > $out/${name}_pmc.log 2>&1
  python scripts/profile_summary.py $out/${name}_trace $out/${name}_pmc $out/profiles/${tag}_${name}.json --command "$*" $alg > $out/${name}_summary.txt
  cat $out/${name}_summary.txt
  cp "$(find $out/${name}_trace -name '*kernel_stats.csv' | head -n 1)" $out/profiles/${tag}_${name}_kernel_stats.csv
  rm -rf $out/${name}_trace $out/${name}_pmc   # (gpurun_out/ travels back only while it is small)
  echo "[profiles] $name done"
}
short="--steps 500 --warmup 100 --cpu-steps 0 --no-second-dtype --no-secondary --repeats 1 --min-warmup-ms 0"
# algorithmic bytes per launch: n (2 * 14 * s + 13 + 4 * nbar), nbar = 24.61 at skin 0.9 (bench.py prints it)
if [ "$part" != rest ]; then
one md_12kbp_f32 "--alg-bytes md_step_kernel<float=5362560" python bench.py $short
one md_12kbp_f64 "--alg-bytes md_step_kernel<double=8050560" python bench.py $short --dtype f64
# the reference's run: every step's positions + quaternions stored by the plain instantiation (+ 7 s N bytes per launch)
one md_12kbp_f32_traj "--alg-bytes md_step_kernel<float=6034560" python bench.py $short --save-every 1
one md_100kbp_f32 "" python bench.py --bp 100000 --steps 150 --warmup 30 --cpu-steps 0 --no-second-dtype --no-secondary --repeats 1 --min-warmup-ms 0
one md_rna2 "" python scripts/bench_rna2.py 500
one md_na1 "" python scripts/bench_na1.py
fi
[ "$part" = md ] && exit 0
# configs[1], 1 kbp: the 16-lanes-per-nucleotide instantiation of the step kernel (2 000 nt <= 6 144)
one md_1kbp_f32 "" python bench.py --bp 1000 $short
one martini_md "" python bench.py --workload martini-bilayer --steps 500 --warmup 100 --cpu-steps 0 --repeats 1
one energy_difftre "" python scripts/bench_energy.py --difftre
one energy_difftre_obs "" python scripts/bench_energy.py --obs
one observables "" python scripts/bench_observables.py
# the headline kernel: trace of the DEFAULT bench command (what the driver's number comes from) + HBM traffic passes
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --cpu-steps 0 --no-secondary > $out/bench_profiled.log 2>&1
cp "$(find $out/stats -name '*kernel_stats.csv' | head -n 1)" $out/profiles/${tag}_bench_kernel_stats.csv
grep '^{' $out/bench_profiled.log | tail -n 1 > $out/profiles/${tag}_bench_profiled_line.json || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py $short > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py $short > $out/pmc_write.log 2>&1
rm -rf $out/stats
python scripts/collect_traffic.py $out/pmc_fetch $out/pmc_write --n 24000 --kernel 'md_step_kernel<float' --out $out/profiles/traffic.json
rm -rf $out/pmc_fetch $out/pmc_write
# ... and for the fp64 kernel, the reference's precision (there was no FETCH / WRITE pass for it)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py $short --dtype f64 > $out/pmc_fetch64.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py $short --dtype f64 > $out/pmc_write64.log 2>&1
python scripts/collect_traffic.py $out/pmc_fetch $out/pmc_write --n 24000 --dtype f64 --kernel 'md_step_kernel<double' --out $out/profiles/traffic_f64.json
rm -rf $out/pmc_fetch $out/pmc_write
echo "[profiles] written to $out/profiles"
