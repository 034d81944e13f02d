"""The N>1 host path (fastf_amd.dist.ShardedPass + HipStages) with real device stages: world_size ranks that share
the box's one GPU, collectives over gloo staged through host memory (RCCL refuses two ranks on one device), result
of the whole job against the oracle.  The same code runs over RCCL with one GPU per rank (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from helpers import Case

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, case_kw, steps, q):
    import torch
    import torch.distributed as dist
    import fastf_amd as F
    from fastf_amd.dist import HipStages, ShardedPass, owner_of_cell
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        case = Case(**case_kw)
        lists = case.lists()
        cbk, gxk, umi, meta = case.packed(lists)
        n = case.n
        cuts = [n * i // world for i in range(world + 1)]
        cuts[1] = int(cuts[1] * 0.7)                      # uneven slices
        a, b = cuts[rank], cuts[rank + 1]
        t = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).to(dev)
        draws = t(F.mt_draws(case.seed, lists.mt_skip, n))
        eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, n_shards=world, shard_rank=rank, device=0)
        eng.reserve(b - a, n)
        sp = ShardedPass(HipStages(eng, dev), max(b - a, 1), dev)
        assert sp.host_staged
        sl = [t(x[a:b].copy()) for x in (cbk, gxk, umi, meta)]
        for _ in range(steps):                            # buffers are reused across steps, as in bench.py
            sp.run(sl[0], sl[1], sl[2], sl[3], b - a, draws)
        f, c, k = sp.gather_coo()
        lf, lc, _ = sp.local_coo()
        owners_ok = bool((owner_of_cell(lc, world) == rank).all()) if len(lc) else True
        q.put((rank, f, c, k, sp.global_counters(), owners_ok, eng.dev_error_bits()))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kw", [
    (2, dict(n=300_000, n_bar=900, n_gene=400, rate_cell=0.8, rate_depth=0.6, umi_pool=512, cell_dist="lognormal",
             p_no_cb=0.03, p_unlisted_cb=0.1, p_bad_xf=0.1, p_n_umi=0.01)),
    (4, dict(n=200_000, n_bar=300, n_gene=100, rate_cell=1.0, rate_depth=1.0, umi_pool=64)),
])
def test_sharded_pass_ranks_share_one_gpu(world, kw):
    ora = Case(**kw).oracle()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kw, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, f, c, k, counters, owners_ok, err in res:
        assert owners_ok and err == 0
        np.testing.assert_array_equal(f, ora["feature"].astype(np.int64))
        np.testing.assert_array_equal(c, ora["cell"].astype(np.int64))
        np.testing.assert_array_equal(k, ora["count"].astype(np.int64))
        hits, sampled, valid, _ = counters
        assert (sampled, valid) == (ora["sampled"], ora["valid"])
