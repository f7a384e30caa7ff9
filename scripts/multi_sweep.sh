#!/bin/bash
# end-to-end bench for several register-tile widths (run from the repo root on the GPU box)
for r in "$@"; do
  echo "== multi_r=$r"
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --no-variants --option multi_r=$r | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('shots/s %.0f  ms/step %.2f  breakdown %s' % (d['value'], d['ms_per_step'], d['breakdown_ms']))
for k,v in d['kernels'].items(): print('   ', k, v)
" || exit 1
done
