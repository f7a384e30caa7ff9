#!/bin/bash
# GPU box, from the repo root: kernel stats + PMC traffic of the general-circuit passes (MODE 1 / MODE 0)
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_modes_$TAG -o run -- python3 $ROOT/scripts/mode1_pass.py 28 > $OUT/prof_modes_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_modes_$TAG -o run -- python3 $ROOT/scripts/mode1_pass.py 28 > $OUT/pmc_fetch_modes_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_modes_$TAG -o run -- python3 $ROOT/scripts/mode1_pass.py 28 > $OUT/pmc_write_modes_$TAG.log 2>&1 || exit 1
echo "modes done"
