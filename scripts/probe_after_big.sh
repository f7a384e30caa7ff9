#!/bin/bash
# does the device stay visible right after a process that held 256 GiB of HBM exits?
cd "$(dirname "$0")/.."
python bench.py --no-cpu --no-variants --steps 2 --warmup 1 > gpurun_out/probe_bench.json 2> gpurun_out/probe_bench.err
echo "bench rc=$?" > gpurun_out/probe.log
for i in $(seq 1 15); do
  t=$(date +%s.%N)
  python -c "
from qcmrf_amd import _lib
try:
    print('count', _lib.device_count(), _lib.device_memory(0))
except Exception as e:
    print('ERR', e)
" >> gpurun_out/probe.log 2>&1
  echo "  at $t" >> gpurun_out/probe.log
  sleep 1
done
