"""diagnostic: where fastf_engine_finish spends its time on the SURVEY 8d device-path of the 200 M-record job (FASTF_PROFILE laps)"""
import os, sys, time
os.environ["FASTF_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastf_amd as F
from fastf_amd import workload
N = 200_000_000
dev = torch.device("cuda", 0)
job = workload.C3(N); lists = job.lists
pb = F.PinnedBatch(N); off = 0
for s in range(workload.SEGMENTS):
    c, g, u, m = job.segment_packed(s, dev); pb.fill(off, c, g, u, m); off += job.seg_len
del c, g, u, m; job._pool = None; torch.cuda.empty_cache()
eng = F.Engine.from_lists(lists, rate_depth=workload.RATE_DEPTH, seed=workload.SEED, umi_max_bases=workload.UMI_LEN, batch_records=8 << 20, key_capacity=N // 4)
for rep in range(4):
    eng.reset(); eng.reseed(workload.SEED, lists.mt_skip); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.push_pinned(pb); t1 = time.perf_counter()
    torch.cuda.synchronize(); t1b = time.perf_counter()
    res = eng.finish(); t2 = time.perf_counter()
    print("rep %d: push %.4f s, device drained after %.4f s more, finish %.4f s, rows %d" % (rep, t1 - t0, t1b - t1, t2 - t1b, res["nnz"]), file=sys.stderr, flush=True)
eng.close(); pb.close()
