#!/bin/bash
# Development aid, ON THE GPU BOX: bash tests/dev/r6/ab.sh <rounds> <variant> <variant> ... - the headline step of bench.py with each variant in turn; a variant is
# a library path relative to the repo, optionally followed by ,ENV=VALUE,... (e.g. mtr_amd/libmtr_hip.so,MTR_SERVICE_WPC=8),
# on ONE box (boxes differ by 1-2 %, which is what a service-kernel change is worth): ms a step per build and round.
R=${1:-3}; shift
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $R); do
  for v in "$@"; do
    lib=${v%%,*}; envs=$(echo "$v" | tr ',' '\n' | tail -n +2 | tr '\n' ' ')
    env $envs MTR_LIB=$GRAFT_REPO_ROOT/$lib python3 bench.py --steps 40 --warmup 4 --cpu-sample 0 --no-latency --no-cli --no-secondary --no-upload-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v round $r: %.2f ms a step, %.0f reads/s, lone launch %.2f ms' % (d['ms_per_step'], d['value'], d['kernels_ms_alone']['launch']))"
  done
done
