"""diagnostic: (cell, feature) group sizes of the configs[2] workload's keys — what K3's dedup has to deal with"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import fastf_amd as F
from fastf_amd import workload
from fastf_amd.dist import HipStages, ShardedPass
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda", 0)
job = workload.C3(N); lists = job.lists
segs = [job.segment_packed(s, dev) for s in range(workload.SEGMENTS)]
cb, gx, umi, meta = (torch.cat([s[i] for s in segs]) for i in range(4)); del segs
draws = torch.from_numpy(F.mt_draws(workload.SEED, lists.mt_skip, N).view(np.int32)).to(dev)
eng = F.Engine.from_lists(lists, rate_depth=workload.RATE_DEPTH, seed=workload.SEED, umi_max_bases=workload.UMI_LEN); eng.reserve(N, N)
sp = ShardedPass(HipStages(eng, dev), N, dev)
sp.run(cb, gx, umi, meta, N, draws); sp.ensure_exact(); torch.cuda.synchronize()
K = int(sp.d_n.item())
keys = sp.sorted[:K]
fs = 1 + 2 * workload.UMI_LEN + 2
grp = keys >> fs
_, cnt = torch.unique_consecutive(grp, return_counts=True)
nonnull = ((keys >> (fs - 1)) & 1) == 1
dup_adj = torch.zeros(K, dtype=torch.bool, device=dev); dup_adj[1:] = (keys[1:] == keys[:-1])
size_of_key = torch.repeat_interleave(cnt, cnt)
print("keys %d groups %d mean %.2f max %d" % (K, cnt.numel(), K / cnt.numel(), int(cnt.max())))
for lim in (1, 2, 4, 8, 16, 32, 64, 2047):
    print("  keys in groups of size <= %4d: %.3f   (groups: %.3f)" % (lim, float((size_of_key <= lim).float().mean()), float((cnt <= lim).float().mean())))
probed = nonnull & (size_of_key > 1) & ~dup_adj
print("non-NULL %.3f, adjacent copies %.3f, probed by the hash set %.3f (of them in groups <= 8: %.3f, <= 16: %.3f)" % (
    float(nonnull.float().mean()), float(dup_adj.float().mean()), float(probed.float().mean()),
    float((probed & (size_of_key <= 8)).float().sum() / probed.float().sum()), float((probed & (size_of_key <= 16)).float().sum() / probed.float().sum())))
print("groups > 2047 keys: %d" % int((cnt > 2047).sum()))
eng.close()
