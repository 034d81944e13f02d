#!/bin/bash
# A/B of library builds on one box: tools/ab_k.sh <rounds> <name> [<name> ...]   (main = fastf_amd/lib, else build/<name>/)
# alternates the builds <rounds> times (bench.py, kernels only) and prints min / median of the step and of every kernel
rounds=$1; shift
out=gpurun_out/r4/ab_$(echo "$@" | tr ' ' '_').txt
mkdir -p gpurun_out/r4; : > $out.raw
for r in $(seq $rounds); do for v in "$@"; do
  if [ $v = main ]; then lib=""; else lib="FASTF_LIB_OVERRIDE=$PWD/build/$v/libfastf_amd.so"; fi
  env $lib python3 bench.py --steps 30 --no-e2e --no-cpu --no-devpath 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'v':'$v','step':d['ms_per_step'],'ok':d['counters']['same_as_single_gpu_reference_run'],'k':{k.split()[0]:v['avg_ms'] for k,v in d['kernels'].items()}}))" >> $out.raw
done; done
python3 - $out.raw <<'PY' | tee $out
import json,sys,statistics as st,collections
rows=[json.loads(l) for l in open(sys.argv[1])]
by=collections.defaultdict(list)
for r in rows: by[r['v']].append(r)
for v,rs in by.items():
    print(v, 'ok' if all(r['ok'] for r in rs) else 'MISMATCH', 'step min %.4f med %.4f' % (min(r['step'] for r in rs), st.median(r['step'] for r in rs)),
          ' '.join('%s %.4f/%.4f' % (k, min(r['k'][k] for r in rs), st.median(r['k'][k] for r in rs)) for k in rs[0]['k']))
PY
