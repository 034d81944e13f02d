"""diagnostic: phase breakdown of filter_pack_kernel (K1b) on the configs[2] shape from s_memtime stamps
(build first: tools/build_variant.sh stamps -DFASTF_STAMPS)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FASTF_LIB_OVERRIDE"] = os.path.join(ROOT, "build", "stamps", "libfastf_amd.so")
import numpy as np, torch
import fastf_amd as F
from fastf_amd import workload, _lib
N = 40_000_000
dev = torch.device("cuda")
job = workload.C3(N)
lists = job.lists
parts = [job.segment_packed(s, dev) for s in range(workload.SEGMENTS)]
d = [torch.cat([p[i] for p in parts]) for i in range(4)]
draws = torch.from_numpy(F.mt_draws(926, lists.mt_skip, N).view(np.int32)).to(dev)
eng = F.Engine.from_lists(lists, rate_depth=0.5, umi_max_bases=12); eng.reserve(N, N)
keys = torch.empty(N, dtype=torch.int64, device=dev); kc = torch.zeros(8, dtype=torch.int64, device=dev); cnt = torch.zeros(4, dtype=torch.int64, device=dev)
T = (N + 4095) // 4096
stamps = torch.zeros(T * 8, dtype=torch.int64, device=dev)
L = _lib.lib(); L.fastf_debug_set_k1_stamps.argtypes = [ctypes.c_void_p]
s = torch.cuda.current_stream().cuda_stream
def run():
    kc.zero_(); cnt.zero_()
    eng.dev_probe_pack(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), N, draws.data_ptr(), N, keys.data_ptr(), N, kc.data_ptr(), cnt.data_ptr(), s)
for _ in range(3): run()
L.fastf_debug_set_k1_stamps(stamps.data_ptr()); run(); torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(T, 8)[: T - 1]
dd = np.diff(st[:, :7], axis=1).astype(np.float64)
names = ["loads+ranks+scan", "tile_base+draw issue", "gene lookup (+draw wait)", "key+slot", "counters+reserve", "key store"]
for i, nm in enumerate(names):
    print("  %-26s median %8.0f  mean %8.0f ticks (100 MHz: x10 ns)" % (nm, np.median(dd[:, i]), dd[:, i].mean()))
tot = (st[:, 6] - st[:, 0]).astype(np.float64)
print("  total per tile             median %8.0f  mean %8.0f" % (np.median(tot), tot.mean()))
print(eng.table_modes, "keys", int(kc[0].item()))
