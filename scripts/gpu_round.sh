#!/bin/bash
# One gpurun call of a development round: GPU tests, the default bench line, step parts, and the
# N = 2 / 4 launches rehearsed as rank processes sharing the box's one GPU.
#   usage (from the repo root, through gpurun): bash scripts/gpu_round.sh <tag> [tests|bench|ranks|parts ...]
set -o pipefail
tag=${1:-dev}; shift
what=${*:-tests bench parts ranks}
out=gpurun_out
mkdir -p $out
for w in $what; do
  case $w in
    tests) timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > $out/${tag}_gputest.log 2>&1; rc=$?; tail -4 $out/${tag}_gputest.log; [ $rc -eq 0 ] || exit $rc ;;
    bench) timeout -k 10 900 python bench.py > $out/${tag}_bench_n1.json 2> $out/${tag}_bench_n1.err; rc=$?; tail -c 600 $out/${tag}_bench_n1.json; [ $rc -eq 0 ] || { tail -20 $out/${tag}_bench_n1.err; exit $rc; } ;;
    parts) timeout -k 10 300 python scripts/time_step_parts.py 1 2 4 > $out/${tag}_step_parts.log 2>&1; rc=$?; cat $out/${tag}_step_parts.log; [ $rc -eq 0 ] || exit $rc ;;
    ranks) for n in 2 4; do
             sleep 8      # the previous launch's shards (and their IPC mappings) are released asynchronously
             timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
               bench.py --gpus $n --steps 6 --warmup 2 > $out/${tag}_bench_${n}ranks_on_1gpu.json 2> $out/${tag}_bench_${n}ranks.err; rc=$?
             tail -c 1500 $out/${tag}_bench_${n}ranks_on_1gpu.json; echo
             [ $rc -eq 0 ] || { tail -30 $out/${tag}_bench_${n}ranks.err; exit $rc; }
           done ;;
    ranks31) for n in 2 4; do
             sleep 8
             timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + n)) \
               bench.py --gpus $n --config 3 --steps 6 --warmup 2 > $out/${tag}_bench_${n}ranks_W31.json 2> $out/${tag}_bench_${n}ranks_W31.err; rc=$?
             tail -c 1500 $out/${tag}_bench_${n}ranks_W31.json; echo
             [ $rc -eq 0 ] || { tail -30 $out/${tag}_bench_${n}ranks_W31.err; exit $rc; }
           done ;;
  esac
done
