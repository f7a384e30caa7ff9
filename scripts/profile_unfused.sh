#!/bin/bash
# GPU box, from the repo root: kernel stats + PMC traffic of the reference's unfused stream (fusion 0, general
# k_multi passes) at 28 qubits.  Summaries: python scripts/pmc_summary.py TAG  ->  profiles/TAG_unfused_*.csv
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_unfused_$TAG -o run -- python3 $ROOT/scripts/trace_passes.py 28 0 > $OUT/prof_unfused_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_unfused_$TAG -o run -- python3 $ROOT/scripts/trace_passes.py 28 0 > $OUT/pmc_fetch_unfused_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_unfused_$TAG -o run -- python3 $ROOT/scripts/trace_passes.py 28 0 > $OUT/pmc_write_unfused_$TAG.log 2>&1 || exit 1
echo "unfused done"
