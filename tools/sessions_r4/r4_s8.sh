#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 1000 python3 -m pytest tests/ -x -q -m gpu -k "not full_size and not adversarial" > gpurun_out/r4/s8_tests.txt 2>&1; rc=$?; grep -q "Memory access fault" gpurun_out/r4/s8_tests.txt && rc=99
tail -15 gpurun_out/r4/s8_tests.txt
[ $rc -ne 0 ] && exit $rc
# the CLI with two aliased devices against the single-device run (same box, Cell-Ranger-shaped 20 M records)
mkdir -p /dev/shm/gb/out && python3 -c "
import sys; sys.path.insert(0,'.')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o build/gen_bam tools/gen_bam.c -lz -lpthread && build/gen_bam /dev/shm/gb/in.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv 20000000 7 12 91 16 3
for v in "" "FASTF_DEVICES=0,0" "" "FASTF_DEVICES=0,0" "FASTF_HOST_DRAWS=1" "FASTF_DEVICES=0,0 FASTF_HOST_DRAWS=1"; do
  s=$(date +%s.%N)
  env FASTF_PROFILE=1 FASTF_HOST_THREADS=16 $v fastf_amd/bin/fastF bam2db -b /dev/shm/gb/in.bam -a /dev/shm/gb/bar.tsv -f /dev/shm/gb/feat.tsv -o /dev/shm/gb/out -c 0.5 -r 0.5 > /dev/null 2> /dev/shm/gb/err.txt
  e=$(date +%s.%N)
  echo "[$v] wall $(echo "$e - $s" | bc) s  md5 $(zcat /dev/shm/gb/out/matrix.mtx.gz | md5sum | cut -c1-12)  $(grep 'phases' /dev/shm/gb/err.txt | cut -c1-200)"
done 2>&1 | tee gpurun_out/r4/s8_multi_cli.txt
rm -rf /dev/shm/gb
