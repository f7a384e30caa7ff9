#!/bin/bash
# GPU box: k_multi plain vs non-temporal by state width (where should multi_nt = -1 switch?), and the
# generator with non-temporal stores
for W in ${QSV_WIDTHS:-20 22 24 25 26 28 30 32}; do
  for nt in 0 1; do
    python bench.py --qubits $W --no-fold --no-cpu --no-variants --steps 10 --warmup 2 --option multi_nt=$nt 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('W=%d multi_nt=$nt' % d['config']['qubits'], {n: (round(x['avg_ms'], 4), round(x['GBps'])) for n, x in d['kernels'].items() if n.startswith('multi')})"
  done
done
for W in 28 29 30 31 32 33 34; do
  for nt in 0 1; do
    python bench.py --qubits $W --no-cpu --no-variants --steps 10 --warmup 2 --option init_prod_nt=$nt 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('generator W=%d nontemporal=$nt' % d['config']['qubits'], {n: (round(x['avg_ms'], 4), round(x['GBps'])) for n, x in d['kernels'].items() if n.startswith('init')})"
  done
done
