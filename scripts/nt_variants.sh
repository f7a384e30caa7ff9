#!/bin/bash
# GPU box: the read+write k_multi pass with non-temporal loads / stores (experiment builds in scripts/_build)
for v in "" NT_LOAD NT_STORE NT_BOTH "" NT_BOTH; do
  lib=""; [ -n "$v" ] && lib=$(pwd)/scripts/_build/libqsv_$v.so
  for cfg in "--config 2" ""; do
    QSV_LIBRARY=$lib python bench.py $cfg --no-fold --no-cpu --no-variants --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('%-9s W=%d' % ('${v:-plain}', d['config']['qubits']), {n: (round(x['avg_ms'], 3), round(x['GBps'])) for n, x in k.items() if n.startswith('multi')})"
  done
done
