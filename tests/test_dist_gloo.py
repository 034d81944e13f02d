"""world_size-2 (and 3) run of the sharded host path over gloo on CPU: the orchestration of
fastf_amd.dist (draw-rank base across ranks, cell-hash ownership, the all-to-all, counters
all-reduce, global (cell, feature) merge) against the oracle on the whole job."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fastf_amd as F
from fastf_amd.dist import ShardedPass, owner_of_cell
from helpers import Case


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, case_kw, q, mode="steps"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dist_doubles import NumpyStages, NumpyWideStages
        case = Case(**case_kw)
        lists = case.lists()
        if mode == "wide":
            cbk, gxk, umi, meta, ext = case.packed_long(lists)
        else:
            cbk, gxk, umi, meta = case.packed(lists)
        n = case.n
        # contiguous slices of the record stream, deliberately uneven
        cuts = [0] + [int(n * (i + 1) / world * (0.8 if i % 2 == 0 and i + 1 < world else 1.0)) for i in range(world)]
        cuts[-1] = n
        a, b = cuts[rank], cuts[rank + 1]
        draws = torch.from_numpy(F.mt_draws(case.seed, lists.mt_skip, n).view(np.int32))
        st = (NumpyWideStages if mode == "wide" else NumpyStages)(lists.cell_keys, lists.feature_keys, F.draw_threshold(case.rate_depth))
        sp = ShardedPass(st, max(b - a, 1), torch.device("cpu"))
        t = lambda x: torch.from_numpy(x[a:b].copy().view(np.int64) if x.dtype == np.uint64 else x[a:b].copy().view(np.int32))
        d = (t(cbk), t(gxk), t(umi), t(meta))
        info = {}
        if mode == "wide":
            for _ in range(2):                                        # (a second step on the same buffers)
                sp.run(*d, b - a, draws, umi_ext=t(ext))
            f, c, k = sp.gather_coo()
        elif mode == "overflow":
            os.environ["FASTF_DIST_CAP_SLACK"] = "0"
            # the capacity of the fixed-size rows is learned from a step over a tenth of the slice; the full slice then
            # outgrows it: ensure_exact() (behind every result) must notice and repeat the step the exact way
            m = max((b - a) // 10, 1)
            sp.run(*(x[:m] for x in d), m, draws)
            sp.ensure_exact()
            small_cap = sp.cap
            sp.run(*d, b - a, draws)
            info["fixed_then_redone"] = bool(sp._fixed_step)          # before the results are asked for
            f, c, k = sp.gather_coo()
            info["cap_grew"] = sp.cap is not None and sp.cap >= small_cap
            info["redone_exactly"] = not sp._fixed_step
        else:
            sp.run(*d, b - a, draws)                                  # learns the capacity (exact protocol)
            sp.ensure_exact()
            per_step = []
            for _ in range(3):
                c0 = sp.n_collectives
                sp.run(*d, b - a, draws)
                per_step.append(sp.n_collectives - c0)
            info["collectives_per_step"] = per_step
            info["fixed"] = bool(sp._fixed_step)
            f, c, k = sp.gather_coo()
        lf, lc, _ = sp.local_coo()
        owners_ok = bool((owner_of_cell(lc, world) == rank).all()) if len(lc) else True
        q.put((rank, f, c, k, sp.global_counters(), owners_ok, info))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "steps"), (3, "steps"), (2, "overflow"), (2, "wide"), (3, "wide")])
def test_sharded_pass_over_gloo(world, mode):
    """the first step learns the per-destination key counts the exact way; every later step is the fixed-capacity form:
    TWO collectives (hit counts, keys), nothing on the host that waits for the device — and the oracle's rows"""
    case_kw = dict(n=6000, n_bar=60, n_gene=25, rate_cell=0.7, rate_depth=0.6, umi_pool=40,
                   p_no_cb=0.05, p_unlisted_cb=0.1, p_bad_xf=0.1, p_n_umi=0.05, p_no_ub=0.02)
    if mode == "wide":                       # 20-base UMIs: the key travels as (group word, value) through two exchanges
        case_kw["umi_len"] = 20
    ora = Case(**case_kw).oracle()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case_kw, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, f, c, k, counters, owners_ok, info in res:
        assert owners_ok
        if mode == "steps":
            assert info["fixed"] and info["collectives_per_step"] == [2, 2, 2], info
        elif mode == "overflow":
            assert info["fixed_then_redone"] and info["redone_exactly"] and info["cap_grew"], info
        np.testing.assert_array_equal(f, ora["feature"].astype(np.int64))
        np.testing.assert_array_equal(c, ora["cell"].astype(np.int64))
        np.testing.assert_array_equal(k, ora["count"].astype(np.int64))
        hits, sampled, valid, _ = counters
        assert (sampled, valid) == (ora["sampled"], ora["valid"])


def test_owner_function_is_balanced_and_stable():
    cells = np.arange(1, 100001)
    for g in (2, 4, 8):
        o = owner_of_cell(cells, g)
        cnt = np.bincount(o, minlength=g)
        assert cnt.min() > 0.9 * len(cells) / g
    assert owner_of_cell(np.array([1, 2, 3, 1000]), 8).tolist() == owner_of_cell(np.array([1, 2, 3, 1000]), 8).tolist()
