"""The N>1 host path (fastf_amd.dist.ShardedPass + HipStages) with real device stages: world_size ranks that share
the box's one GPU, collectives over gloo staged through host memory (RCCL refuses two ranks on one device), result
of the whole job against the oracle.  The same code runs over RCCL with one GPU per rank (bench.py --gpus N)."""
import os
import socket

import numpy as np

from fastf_amd import hostmem
import pytest
import torch.multiprocessing as mp

from helpers import Case

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, case_kw, steps, q, umi_max=12):
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
        ext = None
        if umi_max > 16:
            cbk, gxk, umi, meta, ext = case.packed_long(lists)
        else:
            cbk, gxk, umi, meta = case.packed(lists)
        n = case.n
        cuts = [n * i // world for i in range(world + 1)]
        cuts[1] = int(cuts[1] * 0.7)                      # uneven slices
        a, b = cuts[rank], cuts[rank + 1]
        t = lambda x: hostmem.to_device(x, dev)
        draws = t(F.mt_draws(case.seed, lists.mt_skip, n))
        eng = F.Engine.from_lists(lists, rate_depth=case.rate_depth, seed=case.seed, n_shards=world, shard_rank=rank, device=0,
                                  umi_max_bases=umi_max)
        eng.reserve(b - a, n)
        sp = ShardedPass(HipStages(eng, dev), max(b - a, 1), dev)
        assert sp.host_staged
        sl = [t(x[a:b].copy()) for x in (cbk, gxk, umi, meta)]
        if eng.wide:
            # keys wider than 64 bits: (group word, value) pairs through the exchange, the receiver's engine sorts and reduces them
            dr = sp.prepare_draws(draws)
            for step in range(steps):
                sp.run(sl[0], sl[1], sl[2], sl[3], b - a, draws if step == 0 else dr, umi_ext=None if ext is None else t(ext[a:b].copy()))
            f, c, k = sp.gather_coo()
            lf, lc, _ = sp.local_coo()
            owners_ok = bool((owner_of_cell(lc, world) == rank).all()) if len(lc) else True
            q.put((rank, f, c, k, sp.global_counters(), owners_ok, 0))
            eng.close()
            return
        # buffers are reused across steps, as in bench.py: the first step from the SoA arrays and the 32-bit draws, the rest
        # the way bench.py --gpus N runs them — blocked records, the decision bits made once
        blk = sp.st.block(sl[1], sl[2], sl[3], b - a) if b > a else None
        dr = sp.prepare_draws(draws)
        for step in range(steps):
            if step == 0 or blk is None:
                sp.run(sl[0], sl[1], sl[2], sl[3], b - a, draws)
            else:
                sp.run(sl[0], blk, None, None, b - a, dr)
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
    procs = [ctx.Process(target=_worker, args=(r, world, port, kw, 3, q)) for r in range(world)]     # one exact step, two fixed-capacity ones
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


@pytest.mark.parametrize("world,umi_max,kw", [
    (2, 16, dict(n=150_000, n_bar=210_000, n_gene=70_000, rate_depth=0.9, umi_len=16, umi_pool=4096, p_n_umi=0.01)),       # 71 key bits
    (3, 24, dict(n=200_000, n_bar=2000, n_gene=900, rate_cell=0.7, rate_depth=0.8, umi_len=20, dup_factor=3.0, p_no_cb=0.03,
                 p_unlisted_cb=0.05, p_bad_xf=0.1, p_n_umi=0.02)),
    (2, 32, dict(n=120_000, n_bar=300, n_gene=100, umi_len=30, umi_pool=256, rate_depth=0.7, p_n_umi=0.05)),
])
def test_sharded_pass_with_keys_wider_than_64_bits(world, umi_max, kw):
    """what the single-GPU engine takes, the sharded host path takes too (hashtable.c:70-115, bam2db_ds.c:417-419: any list size,
    any UMI length): fastf_dev_probe_pack_wide -> two exchanges -> fastf_dev_adopt_wide + fastf_engine_finish per rank"""
    ora = Case(**kw).oracle()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kw, 2, q, umi_max)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, f, c, k, counters, owners_ok, err in res:
        assert owners_ok
        np.testing.assert_array_equal(f, ora["feature"].astype(np.int64))
        np.testing.assert_array_equal(c, ora["cell"].astype(np.int64))
        np.testing.assert_array_equal(k, ora["count"].astype(np.int64))
        hits, sampled, valid, errbits = counters
        assert (sampled, valid, errbits) == (ora["sampled"], ora["valid"], 0)


@pytest.mark.parametrize("fixed", [False, True], ids=["exact_protocol", "fixed_capacity_rows"])
def test_pipelined_pass_with_asynchronous_loopback_exchange(fixed):
    """The two-stream pipeline of ShardedPass under real asynchrony: one process plays shard 0 of 2, the collectives
    are device-side copies on the current stream (nothing synchronises the host but the count read-back — and in the
    fixed-capacity form, after the first step, nothing at all), and the input alternates between two different record slices
    from step to step — a shard buffer or counter slot reused too early shows up as a wrong matrix.  Each step's rows
    are checked against the oracle restricted to shard 0's cells."""
    import torch
    import fastf_amd as F
    from fastf_amd.dist import HipStages, ShardedPass, owner_of_cell

    class Loopback(ShardedPass):
        def _all_gather(self, out, inp):
            self.n_collectives += 1
            out.zero_(); out[self.rank:self.rank + 1].copy_(inp)          # the other shard holds no records

        def _all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
            self.n_collectives += 1
            if out.numel() == self.G:
                out.zero_(); out[:1].copy_(inp[:1])                       # only my own shard-0 count comes back
            else:                                                         # the key rows: mine comes back, the other shard's is empty
                row = out.numel() // self.G
                out.zero_(); out[:row].copy_(inp[:row])

        def _all_reduce(self, t, op=None):
            self.n_collectives += 1                                       # (the other shard has nothing to add)

        def _exchange_keys(self, send, recv):
            self.recv[:send[0]].copy_(self.keys_out[0, :send[0]])

        def _gather_small(self, out_cpu, inp_cpu):
            out_cpu.zero_(); out_cpu[self.rank] = inp_cpu[0]

        def _exchange_small(self, out_cpu, inp_cpu):
            out_cpu.zero_(); out_cpu[0] = inp_cpu[0]

    dev = torch.device("cuda", 0)
    kws = [dict(n=400_000, n_bar=600, n_gene=300, rate_cell=0.9, rate_depth=0.7, umi_pool=1024, cell_dist="lognormal",
                p_unlisted_cb=0.1, p_bad_xf=0.1, data_seed=s) for s in (21, 22)]
    cases = [Case(**kw) for kw in kws]
    # the same lists for both slices (same list seed is part of data_seed): rebuild the second case on the first's lists
    cases[1].bt, cases[1].ft, cases[1].bar, cases[1].genes = cases[0].bt, cases[0].ft, cases[0].bar, cases[0].genes
    from fastf_amd import synth
    fl, xf, cb, gx, ub = synth.make_records(cases[1].n, cases[0].bar, cases[0].genes, seed=22, umi_pool=1024,
                                            cell_dist="lognormal", p_unlisted_cb=0.1, p_bad_xf=0.1)
    cases[1].flags, cases[1].xf = fl, xf
    cases[1].cb, cases[1].gx, cases[1].ub = synth.as_cstr(cb), synth.as_cstr(gx), synth.as_cstr(ub)
    lists = cases[0].lists()
    oras = [c.oracle() for c in cases]
    t = lambda x: hostmem.to_device(x, dev)
    packed = [[t(a) for a in c.packed(lists)] for c in cases]
    n = cases[0].n
    draws = t(F.mt_draws(cases[0].seed, lists.mt_skip, n))
    eng = F.Engine.from_lists(lists, rate_depth=cases[0].rate_depth, seed=cases[0].seed, n_shards=2, shard_rank=0, device=0)
    eng.reserve(n, n)
    try:
        sp = Loopback(HipStages(eng, dev), n, dev, world=2, rank=0)
        sp.fixed = fixed
        assert sp.pipelined
        for step in range(12):
            k = step % 2
            c0 = sp.n_collectives
            sp.run(packed[k][0], packed[k][1], packed[k][2], packed[k][3], n, draws)
            if fixed and step >= 1:
                assert sp._fixed_step and sp.n_collectives - c0 == 2          # the hit counts and the keys, nothing else
            if step in (5, 10, 11):                                   # mid-stream and final results
                f, c, cnt = sp.local_coo()
                ora = oras[k]
                mine = owner_of_cell(ora["cell"].astype(np.int64), 2) == 0
                np.testing.assert_array_equal(c, ora["cell"].astype(np.int64)[mine])
                np.testing.assert_array_equal(f, ora["feature"].astype(np.int64)[mine])
                np.testing.assert_array_equal(cnt, ora["count"].astype(np.int64)[mine])
        assert eng.dev_error_bits() == 0
    finally:
        eng.close()


def _rccl_worker(port, q):
    import torch
    import torch.distributed as dist
    from fastf_amd.dist import ShardedPass
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sp = ShardedPass(object(), 64, dev, world=1, rank=0)
        assert sp._nccl and not sp.host_staged
        inp = torch.arange(1000, dtype=torch.int64, device=dev) * 3 + 1
        out = torch.zeros(1000, dtype=torch.int64, device=dev)
        sp._all_to_all_single(out, inp, [1000], [1000])                 # the key exchange's call: int64 keys, split lists
        ok = bool(torch.equal(out, inp))
        sp._all_to_all_single(out[:0], inp[:0], [0], [0])               # a shard that sends and receives nothing
        cnt_in = torch.tensor([17], dtype=torch.int64, device=dev); cnt_out = torch.zeros(1, dtype=torch.int64, device=dev)
        sp._all_to_all_single(cnt_out, cnt_in)                          # the count exchange
        hits = torch.tensor([5], dtype=torch.int64, device=dev); allh = torch.zeros(1, dtype=torch.int64, device=dev)
        sp._all_gather(allh, hits)                                      # the draw-rank base
        red = torch.tensor([3, 4, 5, 0], dtype=torch.int64, device=dev)
        sp._all_reduce(red)                                             # the three counters + error bits
        torch.cuda.synchronize()
        q.put((ok, int(cnt_out.item()), int(allh.item()), red.tolist()))
    finally:
        dist.destroy_process_group()


def test_collectives_on_rccl_with_one_rank():
    """Every collective the sharded pass issues, through torch.distributed's nccl backend (= RCCL) on the box's GPU with a
    world of one rank — all a one-GPU box allows: the calls, dtypes and split lists are the ones the 8-GPU run makes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert res == (True, 17, 5, [3, 4, 5, 0])
