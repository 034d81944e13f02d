"""diagnostic: step time of the sharded pass on ONE GPU with a loopback exchange (one process plays shard 0 of G, the
collectives are device-side copies), pipelined over two streams vs on one stream — shows what the K1 / sort overlap buys"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastf_amd as F
from fastf_amd import synth
from fastf_amd.dist import HipStages, ShardedPass

class Loopback(ShardedPass):
    def _all_gather(self, out, inp): out.zero_(); out[self.rank:self.rank + 1].copy_(inp)
    def _all_to_all_single(self, out, inp, out_splits=None, in_splits=None): out.copy_(inp)     # every "peer" sends me what I send it
    def _gather_small(self, out_cpu, inp_cpu): out_cpu.zero_(); out_cpu[self.rank] = inp_cpu[0]
    def _exchange_small(self, out_cpu, inp_cpu): out_cpu.copy_(inp_cpu)
    def _exchange_keys(self, send, recv):
        o = 0
        for g in range(self.G):                                                                   # G copies: as many keys come back as went out
            self.recv[o:o + send[g]].copy_(self.keys_out[g, :send[g]]); o += send[g]

N, G = 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 8
bt, ft, bar, genes = synth.make_lists(10000, 30000, seed=4242)
lists = F.Lists(bt, ft, 1.0, 926)
rng = np.random.default_rng(1)
cbk = lists.cell_keys[rng.integers(0, lists.n_cells, N)]; gxk = lists.feature_keys[rng.integers(0, lists.n_features, N)]
umi = rng.integers(0, 1 << 20, N, dtype=np.uint32) << 12; meta = np.full(N, 1 | 2 | 4 | (3 << 4), np.uint32)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a.view(np.int32)).to(dev)
d = [t(x) for x in (cbk, gxk, umi, meta, F.mt_draws(926, 0, N))]
for pipe in ("1", "0"):
    os.environ["FASTF_DIST_PIPELINE"] = pipe
    eng = F.Engine.from_lists(lists, n_shards=G, shard_rank=0, device=0); eng.reserve(N, N)
    sp = Loopback(HipStages(eng, dev), N, dev, world=G, rank=0)
    for _ in range(3): sp.run(d[0], d[1], d[2], d[3], N, d[4]); sp.ensure_exact()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): sp.run(d[0], d[1], d[2], d[3], N, d[4])
    t_host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("G=%d pipelined=%s: %.3f ms per step, host returned after %.3f ms per step (%d keys sorted)" % (G, pipe, dt * 1e3, t_host * 1e3, int(sp.d_n.item())))
    eng.close()
