"""diagnostic: K1a/K1b time with n_shards = 1, 2, 4, 8 on one GPU (HIP-event timing from the engine)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastf_amd as F
from fastf_amd import synth
N = 10_000_000
bt, ft, bar, genes = synth.make_lists(10000, 30000, seed=4242)
lists = F.Lists(bt, ft, 1.0, 926)
rng = np.random.default_rng(1)
cbk = lists.cell_keys[rng.integers(0, lists.n_cells, N)]
gxk = lists.feature_keys[rng.integers(0, lists.n_features, N)]
umi = rng.integers(0, 1 << 20, N, dtype=np.uint32) << 12
meta = np.full(N, 1 | 2 | 4 | (3 << 4), np.uint32)
draws = F.mt_draws(926, 0, N)
dev = torch.device("cuda")
t = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a.view(np.int32)).to(dev)
d = [t(x) for x in (cbk, gxk, umi, meta, draws)]
s = torch.cuda.current_stream().cuda_stream
for G in (1, 2, 4, 8):
    eng = F.Engine.from_lists(lists, n_shards=G, shard_rank=0); eng.reserve(N, N)
    keys = torch.empty((G, N), dtype=torch.int64, device=dev); kc = torch.zeros(8, dtype=torch.int64, device=dev); cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    def run():
        kc.zero_(); cnt.zero_()
        eng.dev_probe_pack(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), N, d[4].data_ptr(), N, keys.data_ptr(), N, kc.data_ptr(), cnt.data_ptr(), s)
    for _ in range(3): run()
    torch.cuda.synchronize(); eng.set_timing(True)
    for _ in range(5): run()
    torch.cuda.synchronize()
    a, na = eng.get_timing(0); b, nb = eng.get_timing(4)
    print("n_shards %d: K1a %.1f us  K1b %.1f us   per-shard keys %s" % (G, 1e3 * a / na, 1e3 * b / nb, kc[:G].tolist()))
    eng.close()
