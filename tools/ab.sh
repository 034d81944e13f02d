#!/bin/bash
# A/B of library builds on one box: tools/ab.sh <rounds> <name1> [<name2> ...]   ("main" = fastf_amd/lib, else build/<name>)
rounds=$1; shift
for r in $(seq 1 $rounds); do for v in "$@"; do
  if [ "$v" = main ]; then unset FASTF_LIB_OVERRIDE; else export FASTF_LIB_OVERRIDE=$PWD/build/$v/libfastf_amd.so; fi
  python3 bench.py --steps 40 --no-e2e --no-cpu --no-devpath ${AB_ARGS} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d['counters']['same_as_single_gpu_reference_run'], [(k.split()[0], round(v['avg_ms'],4), round(v['frac'],3)) for k,v in d['kernels'].items()])"
done; done
