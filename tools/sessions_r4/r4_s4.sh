#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 300 tools/e2e_big.sh 10 > gpurun_out/r4/s4_e2e_big_trace.txt 2>&1; tail -40 gpurun_out/r4/s4_e2e_big_trace.txt
timeout -k 10 600 python3 bench.py --steps 20 --no-cpu --no-devpath > gpurun_out/r4/s4_bench.json 2> gpurun_out/r4/s4_bench.err || { tail -20 gpurun_out/r4/s4_bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r4/s4_bench.json'))
for k,v in d['e2e'].items():
    if isinstance(v,dict):
        print(k, v.get('records'), v.get('bam_bytes'), v.get('same_matrix'))
        for q in ('host_inflate','hybrid_inflate'):
            x=v.get(q,{}); print('   ',q, {a:x.get(a) for a in ('value','seconds','seconds_to_outputs_closed','start_up_s','steady_state_records_per_s','steady_state_s','finish_and_write_s','exit_s','matrix_md5')})
"
