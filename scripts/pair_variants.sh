#!/bin/bash
for v in "$@"; do
  echo "== pair_variant=$v"
  timeout -k 10 200 python bench.py --gates --steps 10 --option pair_variant=$v --option lowt_shuffle=0 | python -c "
import sys,json
row=[]
for l in sys.stdin:
    d=json.loads(l)
    if d['gate'] in ('1q_t00','1q_t03','1q_t08','1q_t12','1q_t16','1q_t20','1q_t24','1q_t27'): row.append('%s:%d'%(d['gate'][3:],d['GBps']))
print('  '.join(row))
" || exit 1
done
