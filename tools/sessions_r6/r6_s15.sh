#!/bin/bash
# round 6, session 15: the 20 M-record Cell-Ranger-shaped file (1.66 GB) under larger reader windows — wall time to process exit
set -o pipefail
O=gpurun_out/r6; mkdir -p $O /dev/shm/gb/out
R=$PWD
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o $R/build/gen_bam $R/tools/gen_bam.c -lz -lpthread
$R/build/gen_bam /dev/shm/gb/in.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv 20000000 7 12 91 16 > /dev/null
ls -la /dev/shm/gb/in.bam
python3 - <<'PY' > gpurun_out/r6/s15_windows.txt
import os, subprocess, time
R=os.getcwd()
def run(env):
    best=None
    for _ in range(3):
        for f in os.listdir("/dev/shm/gb/out"): os.unlink("/dev/shm/gb/out/"+f)
        t0=time.perf_counter()
        p=subprocess.run([R+"/fastf_amd/bin/fastF","bam2db","-b","/dev/shm/gb/in.bam","-a","/dev/shm/gb/bar.tsv","-f","/dev/shm/gb/feat.tsv","-o","/dev/shm/gb/out","-c","0.5","-r","0.5"],
                         env=dict(os.environ, FASTF_HOST_THREADS="16", FASTF_PROFILE="1", FASTF_BAM_PROFILE="1", **env), capture_output=True, text=True)
        w=time.perf_counter()-t0
        ph=[l for l in p.stderr.splitlines() if l.startswith("[bam2db] phases")]
        dv=[l for l in p.stderr.splitlines() if "device inflate:" in l]
        md5=subprocess.run("zcat /dev/shm/gb/out/matrix.mtx.gz | md5sum", shell=True, capture_output=True, text=True).stdout.split()[0]
        if best is None or w<best[0]: best=(w, ph[-1] if ph else "", dv[-1][:120] if dv else "", md5, p.returncode)
    return best
for name, env in [("default", {}), ("window 400 MiB", {"FASTF_BAM_WINDOW": str(400<<20)}), ("window 560 MiB", {"FASTF_BAM_WINDOW": str(560<<20)}),
                  ("window 768 MiB", {"FASTF_BAM_WINDOW": str(768<<20)}), ("window 192 MiB", {"FASTF_BAM_WINDOW": str(192<<20)})]:
    w, ph, dv, md5, rc = run(env)
    print("%-16s rc %d  %.3f s to exit = %.1f M records/s  md5 %s\n    %s\n    %s" % (name, rc, w, 20/w, md5[:8], ph, dv), flush=True)
PY
cat gpurun_out/r6/s15_windows.txt
rm -rf /dev/shm/gb
