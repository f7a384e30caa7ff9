#!/bin/bash
for m in 0 1; do
  echo "== kq_mfma=$m"
  timeout -k 10 200 python bench.py --gates --steps 10 --option kq_mfma=$m | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if d['gate'].startswith('kq'): print('%-16s %7.3f ms %6.0f GB/s'%(d['gate'],d['ms'],d['GBps']))
" || exit 1
done
