#!/bin/bash
# A/B helper: run bench.py against alternative builds of the library (diagnostic)
# usage: tools/ab_lib.sh <libdir> [<libdir> ...]   (each contains libfastf_amd.so)
for d in "$@"; do
  FASTF_LIB_OVERRIDE=$d/libfastf_amd.so python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', round(d['value']/1e9,3), round(d['ms_per_step'],4), round(d['roofline']['frac'],3), {k: round(v,4) for k,v in d['kernels_ms'].items()})"
done
