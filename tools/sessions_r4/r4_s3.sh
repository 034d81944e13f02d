#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
for args in "" "--soa" "" "--soa"; do
  python3 bench.py --steps 40 --no-e2e --no-cpu --no-devpath $args 2> gpurun_out/r4/s3_err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args', d['config'].get('record_layout'), round(d['ms_per_step'],4), d['counters']['same_as_single_gpu_reference_run'], [(k.split()[0], round(v['avg_ms'],4), round(v['frac'],3)) for k,v in d['kernels'].items()])" || { tail -20 gpurun_out/r4/s3_err.txt; exit 1; }
done
