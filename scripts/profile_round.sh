#!/bin/bash
# GPU box, from the repo root:  scripts/profile_round.sh TAG
# bench line, rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
# kernel-trace only -- MI355X_MICROARCH.md, HBM section) of the SAME command; gate micro-benchmark.
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
CMD="bench.py --steps 5 --warmup 2 --no-cpu --no-variants"
cd $ROOT && python3 scripts/mode1_pass.py 20 > /dev/null 2>&1   # (import check before the long runs)
cd /tmp || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o run -- python3 $ROOT/$CMD > $OUT/prof_$TAG.log 2>&1 || exit 1
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_$TAG -o run -- python3 $ROOT/$CMD > $OUT/pmc_fetch_$TAG.log 2>&1 || exit 1
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_$TAG -o run -- python3 $ROOT/$CMD > $OUT/pmc_write_$TAG.log 2>&1 || exit 1
echo "write done"
# the full-width sweep path (fold_fresh off) at the 28-qubit roofline config
CMD2="bench.py --config 2 --steps 5 --warmup 2 --no-cpu --no-variants --no-fold"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_sweeps_$TAG -o run -- python3 $ROOT/$CMD2 > $OUT/prof_sweeps_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_sweeps_$TAG -o run -- python3 $ROOT/$CMD2 > $OUT/pmc_fetch_sweeps_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_sweeps_$TAG -o run -- python3 $ROOT/$CMD2 > $OUT/pmc_write_sweeps_$TAG.log 2>&1 || exit 1
echo "sweeps done"
# ... and at the 34-qubit bench workload
cd /tmp || exit 1
CMD3="bench.py --steps 3 --warmup 1 --no-cpu --no-variants --no-fold"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_sweeps34_$TAG -o run -- python3 $ROOT/$CMD3 > $OUT/prof_sweeps34_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_sweeps34_$TAG -o run -- python3 $ROOT/$CMD3 > $OUT/pmc_fetch_sweeps34_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_sweeps34_$TAG -o run -- python3 $ROOT/$CMD3 > $OUT/pmc_write_sweeps34_$TAG.log 2>&1 || exit 1
echo "sweeps34 done"
# general-circuit passes: MODE 1 (complex tables, R = 5) and MODE 0 (masked ops, R = 4), 28 qubits
cd /tmp || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_modes_$TAG -o run -- python3 $ROOT/scripts/mode1_pass.py 28 > $OUT/prof_modes_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_modes_$TAG -o run -- python3 $ROOT/scripts/mode1_pass.py 28 > $OUT/pmc_fetch_modes_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_modes_$TAG -o run -- python3 $ROOT/scripts/mode1_pass.py 28 > $OUT/pmc_write_modes_$TAG.log 2>&1 || exit 1
echo "modes done"
cd $ROOT
timeout -k 10 300 python bench.py --gates --steps 20 > $OUT/gates_$TAG.jsonl 2> $OUT/gates_$TAG.err || exit 1
echo "gates done"
