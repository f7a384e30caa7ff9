#!/bin/bash
# knob sweep on the gate microbench (subset printed); run from the repo root on the GPU box
for opt in "$@"; do
  echo "== $opt"
  timeout -k 10 200 python bench.py --gates --steps 10 --option $opt | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d['gate'] in ('1q_t00','1q_t03','1q_t08','1q_t12','1q_t20','1q_t27','x_t27','mux_rx_c3_t27','diag3','cp_c12_t27','norm_pass'):
        print('%-16s %7.3f ms %6.0f GB/s'%(d['gate'],d['ms'],d['GBps']))
" || exit 1
done
