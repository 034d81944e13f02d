"""diagnostic: SURVEY 8d device-path (pinned host SoA -> COO on the host) of the 200 M-record job through ONE engine handle
driving G aliased devices (every shard on device 0: what a one-GPU box can run of fastf_engine_config_t.n_devices), next
to the single-device engine on the same pinned records.   python tools/multi_devpath.py [G] [records]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastf_amd as F
from fastf_amd import workload
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000_000
dev = torch.device("cuda", 0)
job = workload.C3(N)
lists = job.lists
pb = F.PinnedBatch(N)
off = 0
for s in range(workload.SEGMENTS):
    c, g, u, m = job.segment_packed(s, dev)
    pb.fill(off, c, g, u, m); off += job.seg_len
del c, g, u, m
job._pool = None
torch.cuda.empty_cache()
res0 = None
for label, kw in (("single", {}), ("%d aliased devices" % G, {"devices": [0] * G}), ("single", {}), ("%d aliased devices" % G, {"devices": [0] * G})):
    eng = F.Engine.from_lists(lists, rate_depth=workload.RATE_DEPTH, seed=workload.SEED, umi_max_bases=workload.UMI_LEN,
                              batch_records=8 << 20, key_capacity=N // 4, **kw)
    best = None
    try:
        for rep in range(3):
            eng.reset(); eng.reseed(workload.SEED, lists.mt_skip)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); eng.push_pinned(pb); t1 = time.perf_counter(); res = eng.finish(); t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]: best = (t2 - t0, t1 - t0, t2 - t1)
        key = (res["total"], res["sampled"], res["valid"], res["nnz"])
        if res0 is None: res0 = key
        extra = " per-device records %s" % eng.device_records(G) if kw else ""
        print("%-20s %.4f s (push %.4f, finish %.4f)  %.2f G records/s  same result %s%s" % (label, best[0], best[1], best[2], N / best[0] / 1e9, key == res0, extra), flush=True)
    finally:
        eng.close()
pb.close()
