#!/bin/bash
for zt in 0 1; do
  echo "== zero_tracking=$zt"
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --no-variants --option zero_tracking=$zt | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('shots/s %.0f  ms/step %.2f  breakdown %s' % (d['value'], d['ms_per_step'], d['breakdown_ms']))
for k,v in d['kernels'].items(): print('   ', k, v)
" || exit 1
done
