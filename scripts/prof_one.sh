#!/bin/bash
# dev: kernel trace + one SQ PMC pass of one command, folded by profile_summary.py:  scripts/prof_one.sh <name> <python args...>
set -eo pipefail
name=$1; shift
out=gpurun_out/prof_$name
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python "$@" > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $out/pmc -- python "$@" > $out/pmc.log 2>&1
python scripts/profile_summary.py $out/trace $out/pmc $out/summary.json --command "python $*"
rm -rf $out/trace $out/pmc
