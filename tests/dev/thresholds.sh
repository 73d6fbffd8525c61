for n in 1000 2000 3000 4000 6000; do
  for m in "staged:MTR_STAGED=1" "perread:MTR_STAGED=0 MTR_SPLIT=0" "split:MTR_STAGED=0 MTR_SPLIT=1"; do
    name=${m%%:*}; envs=${m#*:}
    env $envs python bench.py --reads $n --steps 20 --warmup 2 --cpu-sample 0 --no-latency --no-cli > /tmp/o.json 2>/dev/null
    python3 -c "
import json;d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]);print('$n $name', round(d['value']), 'reads/s', round(d['ms_per_step'],2),'ms/step', 'lone', round(d['kernels_ms_alone']['launch'],2))"
  done
done
