"""diagnostic: phase breakdown of filter_pack_kernel (K1b) from s_memtime stamps (-DFASTF_STAMPS build)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FASTF_LIB_OVERRIDE"] = os.path.join(ROOT, "build", "stamps", "libfastf_amd.so")
import numpy as np, torch
import fastf_amd as F
from fastf_amd import synth, _lib
N = 10_000_000
bt, ft, bar, genes = synth.make_lists(10000, 30000, seed=4242)
lists = F.Lists(bt, ft, 1.0, 926)
rng = np.random.default_rng(1)
cbk = lists.cell_keys[rng.integers(0, lists.n_cells, N)]
gxk = lists.feature_keys[rng.integers(0, lists.n_features, N)]
umi = rng.integers(0, 1 << 20, N, dtype=np.uint32) << 12
meta = np.full(N, 1 | 2 | 4 | (3 << 4), np.uint32)
draws = F.mt_draws(926, 0, N)
eng = F.Engine.from_lists(lists); eng.reserve(N, N)
dev = torch.device("cuda")
t = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a.view(np.int32)).to(dev)
d = [t(x) for x in (cbk, gxk, umi, meta, draws)]
keys = torch.empty(N, dtype=torch.int64, device=dev); kc = torch.zeros(8, dtype=torch.int64, device=dev); cnt = torch.zeros(4, dtype=torch.int64, device=dev)
T = (N + 4095) // 4096
stamps = torch.zeros(T * 8, dtype=torch.int64, device=dev)
L = _lib.lib(); L.fastf_debug_set_k1_stamps.argtypes = [ctypes.c_void_p]
s = torch.cuda.current_stream().cuda_stream
def run():
    kc.zero_(); cnt.zero_()
    eng.dev_probe_pack(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), N, d[4].data_ptr(), N, keys.data_ptr(), N, kc.data_ptr(), cnt.data_ptr(), s)
for _ in range(3): run()
L.fastf_debug_set_k1_stamps(stamps.data_ptr()); run(); torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(T, 8)[: T - 1]
dd = np.diff(st[:, :7], axis=1).astype(np.float64)
names = ["loads+ranks+scan", "tile_base+draw gather", "alive/xf logic", "feature probe", "key+shard+hist", "counters+reserve", "key store+hist flush"]
for i, nm in enumerate(names[:6]):
    print("  %-24s median %8.0f  mean %8.0f cycles" % (nm, np.median(dd[:, i]), dd[:, i].mean()))
tot = (st[:, 6] - st[:, 0]).astype(np.float64)
print("  total per tile           median %8.0f  mean %8.0f" % (np.median(tot), tot.mean()))
