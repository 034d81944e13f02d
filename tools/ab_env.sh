#!/bin/bash
# A/B of run-time knobs on one box: tools/ab_env.sh <rounds> "VAR=a" "VAR=b" ...   (each argument is one environment assignment list)
rounds=$1; shift
for r in $(seq 1 $rounds); do for v in "$@"; do
  env $v python3 bench.py --steps 40 --no-e2e --no-cpu --no-devpath ${AB_ARGS} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d['counters']['same_as_single_gpu_reference_run'], d['config']['radix_passes_executed'], [(k.split()[0], round(v['avg_ms'],4), round(v['frac'],3)) for k,v in d['kernels'].items()])"
done; done
