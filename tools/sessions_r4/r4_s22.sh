#!/bin/bash
mkdir -p gpurun_out/r4
timeout -k 10 200 python3 tools/mt_rate.py 2>&1 | tee gpurun_out/r4/s22_mt_rate.txt
for hd in 0 1; do FASTF_HOST_DRAWS=$hd timeout -k 10 300 python3 bench.py --steps 5 --no-e2e --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('host_draws=$hd', d['device_path']['value'], d['device_path']['h2d_GBs'], d['device_path']['runs_s'])" | tee -a gpurun_out/r4/s22_mt_rate.txt; done
