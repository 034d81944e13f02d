"""diagnostic: device-path step at a configs[2]-like key layout (25 k sampled of 50 k barcodes x 36 601 genes, 12-bp UMIs):
prints key bits, skip bits, passes and per-kernel times for 40 M packed records (keep-all)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastf_amd as F
from fastf_amd import synth
from fastf_amd.dist import HipStages, ShardedPass
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
bt, ft, bar, genes = synth.make_lists(50_000, 36_601, seed=77)
lists = F.Lists(bt, ft, 0.5, 926)
rng = np.random.default_rng(1)
w = rng.lognormal(0, 1, lists.n_cells); w /= w.sum()
gw = 1.0 / np.arange(1, 36_602) ** 1.1; gw /= gw.sum()
n_mol = N // 4
mc = rng.choice(lists.n_cells, size=n_mol, p=w); mg = rng.choice(36_601, size=n_mol, p=gw); mu = rng.integers(0, 1 << 24, size=n_mol, dtype=np.uint32)
src = rng.integers(0, n_mol, size=N)
if os.environ.get("C3_MISSES"):       # the real shape of configs[2]: CB tags of all 50 k barcodes (half of them not sampled), 5 % without CB
    alld = F.Lists(bt, ft, 1.0, 926)
    cbk = alld.cell_keys[rng.integers(0, 50_000, size=N)]
    cbk[rng.random(N) < 0.05] = 0
else:
    cbk = lists.cell_keys[mc[src]]
gxk = lists.feature_keys[mg[src]]; umi = (mu[src] << np.uint32(8)).astype(np.uint32)
meta = np.full(N, 1 | 2 | 4 | (3 << 4), np.uint32)
draws = F.mt_draws(926, lists.mt_skip, N)
dev = torch.device("cuda")
t = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a.view(np.int32)).to(dev)
d = [t(x) for x in (cbk, gxk, umi, meta, draws)]
eng = F.Engine.from_lists(lists, rate_depth=1.0, seed=926, umi_max_bases=12); eng.reserve(N, N)
sp = ShardedPass(HipStages(eng, dev), N, dev)
for _ in range(3):
    sp.run(d[0], d[1], d[2], d[3], N, d[4]); sp.ensure_exact()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): sp.run(d[0], d[1], d[2], d[3], N, d[4])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("key_bits %d skip %d passes %d (skip_low=%s) tables %s" % (eng.key_bits, eng.skip_bits, eng.sort_passes(sp.st.skip_low), sp.st.skip_low, eng.table_modes))
eng.set_timing(True)
for _ in range(3): sp.run(d[0], d[1], d[2], d[3], N, d[4])
torch.cuda.synchronize()
a, na = eng.get_timing(0); b, nb = eng.get_timing(4)
print("K1a %.1f us  K1b %.1f us" % (1e3 * a / na, 1e3 * b / nb))
print("step %.3f ms  %.2f G records/s  rows %d keys %d" % (dt * 1e3, N / dt / 1e9, int(sp.nnz.item()), int(sp.d_n.item())))
