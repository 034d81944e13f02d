#!/bin/bash
# diagnostic: the CLI over and over on one generated Cell-Ranger-shaped BAM — the matrix must be the same bytes every time
# (the threaded front end hands host-packed and device-packed batches over in file order whatever the timing):
#   tools/e2e_soak.sh [records] [seq_len] [runs]
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-20000000}; SL=${2:-91}; RUNS=${3:-25}
mkdir -p /dev/shm/gb/out
python3 -c "
import sys; sys.path.insert(0,'$R')
from fastf_amd import synth
bt,ft,_,_=synth.make_lists(50000,36601,seed=77); open('/dev/shm/gb/bar.tsv','wb').write(bt); open('/dev/shm/gb/feat.tsv','wb').write(ft)"
gcc -O2 -o $R/build/gen_bam $R/tools/gen_bam.c -lz -lpthread
$R/build/gen_bam /dev/shm/gb/in.bam /dev/shm/gb/bar.tsv /dev/shm/gb/feat.tsv $N 7 12 $SL 16 > /dev/null
python3 - "$R" $RUNS <<'PY'
import hashlib, os, subprocess, sys, zlib
R, runs = sys.argv[1], int(sys.argv[2])
envs = [{}, {"FASTF_DEVICES": "0,0"}, {"FASTF_DEVICES": "0,0,0,0"}, {"FASTF_GPU_INFLATE": "0"}, {"FASTF_GPU_PARSE": "0"}, {"FASTF_BATCH_RECORDS": "1000000"}, {"FASTF_BAM_WINDOW": str(48 << 20)}, {"FASTF_HOST_THREADS": "7"}, {"FASTF_HOST_DRAWS": "1"}]
seen = {}
for i in range(runs):
    e = envs[i % len(envs)] if i >= runs // 2 else {}
    for f in os.listdir("/dev/shm/gb/out"): os.unlink("/dev/shm/gb/out/" + f)
    p = subprocess.run([R + "/fastf_amd/bin/fastF", "bam2db", "-b", "/dev/shm/gb/in.bam", "-a", "/dev/shm/gb/bar.tsv", "-f", "/dev/shm/gb/feat.tsv",
                        "-o", "/dev/shm/gb/out", "-c", "0.5", "-r", "0.5", "-u"], env=dict(os.environ, **e), capture_output=True, text=True)
    if p.returncode: print("run", i, e, "rc", p.returncode, p.stderr[-300:]); sys.exit(1)
    h = hashlib.md5()
    for name in ("matrix.mtx.gz", "umi.tsv.gz"):
        h.update(zlib.decompress(open("/dev/shm/gb/out/" + name, "rb").read(), 31) if False else subprocess.run(["zcat", "/dev/shm/gb/out/" + name], capture_output=True).stdout)
    counters = [l for l in p.stdout.splitlines() if "fastQ reads" in l]
    key = (h.hexdigest(), tuple(c.split(":")[-1].strip() for c in counters))
    seen.setdefault(key, []).append((i, tuple(e.items())))
print("runs", runs, "distinct results", len(seen))
for k, v in seen.items(): print(k, len(v), v[:3])
sys.exit(0 if len(seen) == 1 else 2)
PY
rc=$?
rm -rf /dev/shm/gb
exit $rc
