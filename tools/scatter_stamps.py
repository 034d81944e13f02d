"""diagnostic: phase breakdown of scatter_kernel from in-kernel s_memtime stamps (needs the -DFASTF_STAMPS build)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["FASTF_LIB_OVERRIDE"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "stamps", "libfastf_amd.so")
import fastf_amd._lib as _lib
import fastf_amd as F
N = 10_000_000
cells = np.arange(1, 1001, dtype=np.uint64) | (np.uint64(1) << np.uint64(62))
feats = np.arange(1, 501, dtype=np.uint64) | (np.uint64(2) << np.uint64(62))
eng = F.Engine(cells, feats)
rng = np.random.default_rng(1)
keys = rng.integers(0, 1 << 56, size=N, dtype=np.uint64)
dev = torch.device("cuda")
d_keys = torch.from_numpy(keys.view(np.int64)).to(dev); d_tmp = torch.empty_like(d_keys)
d_n = torch.tensor([N], dtype=torch.int64, device=dev)
IPT = int(os.environ.get("FASTF_SORT_IPT", "13")); T = (N + IPT * 512 - 1) // (IPT * 512)
stamps = torch.zeros(T * 8, dtype=torch.int64, device=dev)
L = _lib.lib(); L.fastf_debug_set_stamps.argtypes = [ctypes.c_void_p]
s = torch.cuda.current_stream().cuda_stream
for it in range(3):
    eng.dev_sort(d_keys.data_ptr(), d_tmp.data_ptr(), d_n.data_ptr(), N, key_bits=56, stream=s)
L.fastf_debug_set_stamps(stamps.data_ptr())
eng.dev_sort(d_keys.data_ptr(), d_tmp.data_ptr(), d_n.data_ptr(), N, key_bits=8, stream=s)   # one pass
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(T, 8)[: T - 1]
d = np.diff(st[:, :7], axis=1).astype(np.float64)
names = ["load", "rank", "binstart", "lds_scatter", "writeout_issue", "store_drain"]
print("tiles", len(st), " cycles per phase (median / mean):")
for i, nm in enumerate(names):
    print("  %-15s %8.0f %8.0f" % (nm, np.median(d[:, i]), d[:, i].mean()))
tot = (st[:, 6] - st[:, 0]).astype(np.float64)
print("  total per tile  %8.0f %8.0f" % (np.median(tot), tot.mean()))
t0 = st[:, 0].min(); print("  kernel span (cycles, from stamps):", st[:, 6].max() - t0)
start = np.sort(st[:, 0] - t0)
print("  tile start times: p10 %d p50 %d p90 %d max %d" % tuple(np.percentile(start, [10, 50, 90, 100])))
