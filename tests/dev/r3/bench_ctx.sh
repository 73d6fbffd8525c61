#!/bin/bash
# Development aid: bench.py with 2, 3, 4 contexts keeping launches in flight
for c in ${@:-2 3 4}; do
  MTR_BENCH_CONTEXTS=$c python bench.py --steps 12 --no-cli --no-latency --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contexts $c:', round(d['value']), 'reads/s', round(d['ms_per_step'],2), 'ms/step; launch', round(d['kernels_ms']['launch'],1), 'ms', d['kernels_ms']['phases'])"
done
