#!/bin/bash
# A/B of library builds + run-time knobs on one box: tools/ab_lib.sh <rounds> "<name> [VAR=x ...]" ...   (name "main" = fastf_amd/lib, else build/<name>)
rounds=$1; shift
for r in $(seq 1 $rounds); do for spec in "$@"; do
  set -- $spec; v=$1; shift
  if [ "$v" = main ]; then lib=""; else lib="FASTF_LIB_OVERRIDE=$PWD/build/$v/libfastf_amd.so"; fi
  env $lib "$@" python3 bench.py --steps 40 --no-e2e --no-cpu --no-devpath ${AB_ARGS} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$spec', round(d['ms_per_step'],4), d['counters']['same_as_single_gpu_reference_run'], d['config']['radix_passes_executed'], [(k.split()[0], round(v['avg_ms'],4), round(v['frac'],3)) for k,v in d['kernels'].items()])"
  set -- "$@"
done; done
