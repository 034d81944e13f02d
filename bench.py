#!/usr/bin/env python3
"""bench.py — the bam2db hot path on MI355X, measured on BASELINE.json configs[2].

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (fastf_amd/workload.py): 200 M synthetic records, 50 k barcodes x 36 601 genes,
--cell 0.5 --depth 0.5 --seed 926 (the shape of the reference's own benchmark/fastF_disk.sh).
A step = one pass of the hot path over the whole job: K1 probe/filter/pack -> [all-to-all] -> K2 LSD
radix sort -> K3 segmented unique/reduce -> COO.  With N GPUs the SAME job is split over the ranks
(strong scaling): rank r owns a contiguous slice of the record stream, keys travel to their cell's
owner in one all-to-all.

The one JSON line carries
  value            records/s of the timed steps, inputs resident in HBM when the clock starts
  roofline         the kernel that takes most of the step: algorithmic bytes / its HIP-event time
  whole_path       SURVEY 8d's B and B_read over the step time (nominal and executed radix passes)
  device_path      N=1: pinned host SoA -> hipMemcpyAsync -> kernels -> COO on the host (PCIe-inclusive): SURVEY 8d's
                   device-path scope; its rate is repeated at the top level as device_path_records_per_s
  e2e              N=1: the real CLI on generated BAM files, process start to process EXIT (outputs-closed beside it);
                   the Cell-Ranger-shaped rate is repeated at the top level as e2e_records_per_s_to_process_exit
  (`value` itself is the resident-input rate the bench contract asks for; the two rates above are what a caller of
  bam2db() sees and are never folded into it)
  cpu_baseline     N=1: the CPU oracle on a bounded sample of the same records, 1 core (+ parity check)
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# (no on-the-fly pinning of pageable host memory by the HIP runtime in this process: DESIGN section 14; the transfers of this
#  script go through pinned staging anyway — fastf_amd/hostmem.py — this covers what torch copies by itself)
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1000000")

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before fastf_amd: one shared HIP runtime)
import torch.distributed as dist  # noqa: E402

import fastf_amd as F  # noqa: E402
from fastf_amd import hostmem, workload  # noqa: E402
from fastf_amd.dist import HipStages, ShardedPass  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PCIE_GBS = 55.0                # Gen5 x16, what a pinned hipMemcpyAsync reaches


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_table(eng, N, H, K, Z, draw_bits, steps_timed):
    """per-kernel HIP-event averages (ms) with the bytes each kernel has to MOVE per launch;
    c = bytes per entry of the cell-index scratch between K1a and K1b (2 when n_cells <= 65535, else 4).
    K1b reads ONE BIT per CB hit when the draw stream is the decision stream (the product's form): its bytes are counted
    that way, and SURVEY 8d's figure (4 bytes per hit) is kept beside it as bytes_8d — never as the fraction's numerator."""
    c = eng.cell_scratch_bytes
    draw_bytes = (H + 7) // 8 if draw_bits else 4 * H
    names = {0: ("probe_cells (K1a)", (8 + c) * N, None), 4: ("filter_pack (K1b)", (16 + c) * N + draw_bytes + 8 * K, (16 + c) * N + 4 * H + 8 * K),
             3: ("tile_count (K2, per pass; the first pass has none: K1b left its histograms)", 8 * K, None), 1: ("scatter (K2, per pass)", 16 * K, None),
             2: ("reduce_hashed+giant_groups (K3; the span scan is the last workgroup of giant_groups)", 8 * K + 12 * Z, None),
             5: ("rows_gather (not part of the step: concatenates K3's row regions where the rows are wanted)", 24 * Z, None)}
    out = {}
    for which, (nm, b, b8d) in names.items():
        ms, n = eng.get_timing(which)
        if n:
            avg = ms / n
            out[nm] = {"avg_ms": avg, "launches_timed": int(n), "launches_per_step": n / steps_timed, "bytes_per_launch": b,
                       "GBs": b / (avg * 1e-3) / 1e9, "frac": b / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if b8d is not None:
                out[nm]["bytes_8d"] = b8d
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 100 steps of configs[2] = 0.27 s timed
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["c3", "c2"], default="c3",
                    help="c3 = BASELINE configs[2] (the headline, default); c2 = BASELINE configs[1], 10 M keep-all records (round 1's workload)")
    ap.add_argument("--records", type=int, default=0, help="records of the whole job (all GPUs together); default 200 M (c3) / 10 M (c2)")
    ap.add_argument("--cpu-sample", type=int, default=75_000_000, help="records timed on the CPU oracle (rank 0, N=1): whole segments of the job, about 10 s of one core")
    ap.add_argument("--soa", action="store_true", help="resident inputs as the four SoA arrays instead of the engine's blocked staging layout")
    ap.add_argument("--draw-words", action="store_true", help="hand K1 the 32-bit draws every step (converted to decision bits per step) instead of the bits made once")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-devpath", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--e2e-records", type=str, default="40000000,20000000", help="records of the skinny and the Cell-Ranger-shaped BAM")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FASTF_BENCH_ONE_DEVICE"):      # rehearsal of the N>1 code path on a 1-GPU box
        local = 0
    if args.gpus != world and world == 1 and args.gpus > 1:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("FASTF_BENCH_BACKEND", "nccl")     # "gloo" only for the one-GPU rehearsal
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    G = world
    if workload.SEGMENTS % G:
        sys.exit("the job is cut into %d segments: --gpus must divide it" % workload.SEGMENTS)

    # ---- the job: every rank builds the same lists and molecule pool, then its own slice of the record stream ----
    t_gen = time.perf_counter()
    c2 = args.workload == "c2"
    job = workload.C2(args.records or 10_000_000) if c2 else workload.C3(args.records or 200_000_000)
    rate_cell, rate_depth, umi_bases = (1.0, 1.0, 12) if c2 else (workload.RATE_CELL, workload.RATE_DEPTH, workload.UMI_LEN)
    N_total, seg_per_rank = job.n_total, workload.SEGMENTS // G
    n_local = seg_per_rank * job.seg_len
    lists = job.lists
    cb = torch.empty(n_local, dtype=torch.int64, device=dev); gx = torch.empty_like(cb)
    umi = torch.empty(n_local, dtype=torch.int32, device=dev); meta = torch.empty_like(umi)
    for i in range(seg_per_rank):
        a = i * job.seg_len
        c, g, u, m = job.segment_packed(rank * seg_per_rank + i, dev)
        cb[a:a + job.seg_len], gx[a:a + job.seg_len], umi[a:a + job.seg_len], meta[a:a + job.seg_len] = c, g, u, m
    del c, g, u, m
    job._pool = None
    torch.cuda.empty_cache()
    draws_h = F.mt_draws(workload.SEED, lists.mt_skip, N_total)      # job-wide draw stream: draw i belongs to the i-th CB hit
    d_draws = hostmem.to_device(draws_h, dev)                         # (pinned staging: fastf_amd/hostmem.py)
    torch.cuda.synchronize()
    if rank == 0:
        log("bench: job generated in %.1f s (%d records per rank)" % (time.perf_counter() - t_gen, n_local))

    eng = F.Engine.from_lists(lists, rate_depth=rate_depth, seed=workload.SEED, umi_max_bases=umi_bases,
                              n_shards=G, shard_rank=rank, device=local)
    eng.reserve(n_local, n_local)
    stages = HipStages(eng, dev)
    sp = ShardedPass(stages, n_local, dev)
    # resident inputs in the engine's own staging layout: the cb keys as an array, gx | umi | meta in blocked runs of 256 records
    # (include/fastf_amd.h "BLOCKED record layout"; a host batch gets there by pitched copies); --soa keeps the four arrays
    blk = None if args.soa else stages.block(gx, umi, meta, n_local)
    if blk is not None:
        torch.cuda.synchronize()
        del gx, umi, meta
        gx = umi = meta = None
        torch.cuda.empty_cache()

    # the draw stream as K1b reads it: one decision bit per CB hit (draw < threshold), made once from the resident draws —
    # in the product mt_fill_kernel writes these bits itself (include/fastf_amd.h "The decision stream"); --draw-words hands
    # the 32-bit draws to every step instead (one conversion pass per step)
    d_stream = d_draws if args.draw_words else sp.prepare_draws(d_draws)
    torch.cuda.synchronize()

    def step():
        if blk is not None:
            sp.run(cb, blk, None, None, n_local, d_stream)
        else:
            sp.run(cb, gx, umi, meta, n_local, d_stream)

    for _ in range(max(args.warmup, 1)):
        step()
        sp.ensure_exact()      # data with very deep (cell, feature) groups switches the sort to all digits here, once
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    err = eng.dev_error_bits()
    hits, sampled, valid, _ = sp.global_counters()

    # ---- the step WITH its draw generation (N = 1): the job's draws — one per CB hit — made inside the clock by the device's own
    #      MT19937 (fastf_dev_mt_decisions: sub-streams seated by jump-ahead, generated side by side), then the same step ----
    with_draws = None
    K_ref = int(sp.d_n.item())
    if world == 1 and not args.draw_words:
        n_gen_steps = max(3, min(args.steps, 20))
        bits = stages.mt_decisions(workload.SEED, lists.mt_skip, hits)         # (first call: polynomials up, buffers sized)
        torch.cuda.synchronize()
        tg0 = time.perf_counter()
        for _ in range(n_gen_steps):
            bits = stages.mt_decisions(workload.SEED, lists.mt_skip, hits, out=bits)
        torch.cuda.synchronize()
        t_gen_only = (time.perf_counter() - tg0) / n_gen_steps

        def step_gen():
            b = stages.mt_decisions(workload.SEED, lists.mt_skip, hits, out=bits)
            if blk is not None:
                sp.run(cb, blk, None, None, n_local, b)
            else:
                sp.run(cb, gx, umi, meta, n_local, b)
        step_gen(); torch.cuda.synchronize()
        tg0 = time.perf_counter()
        for _ in range(n_gen_steps):
            step_gen()
        torch.cuda.synchronize()
        t_with = (time.perf_counter() - tg0) / n_gen_steps
        h2, s2, v2, _ = sp.global_counters()
        with_draws = {"ms_per_step": t_with * 1e3, "records_per_s": N_total / t_with, "steps": n_gen_steps,
                      "draw_generation_ms": t_gen_only * 1e3, "draws_per_step": hits, "draws_per_s": hits / t_gen_only,
                      "same_counters_as_the_resident_stream": (h2, s2, v2) == (hits, sampled, valid) and int(sp.d_n.item()) == K_ref,
                      "scope": "fastf_dev_mt_decisions (host seeds and skips the stream, the device seats 624 x 256-draw sub-streams by jump-ahead and generates them side by side, draw_bits packs the decisions; the call synchronises) + the step above"}
    K_local = int(sp.d_n.item())
    Z_local = int(sp.nnz.item())
    tot = torch.tensor([K_local, Z_local], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(tot)
    K_job, Z_job = int(tot[0].item()), int(tot[1].item())

    # ---- per-kernel HIP-event timing (events on the launch stream, a few extra steps after the timed region) ----
    eng.set_timing(True)
    STEPS_TIMED = 3
    for _ in range(STEPS_TIMED):
        step()
        sp.st.rows_gather(sp.d_n, sp.feature, sp.cell, sp.count)      # concatenation of K3's row regions (timed as its own line)
    torch.cuda.synchronize()
    H_local = hits // G
    ktab = kernel_table(eng, n_local, H_local, K_local, Z_local, not args.draw_words, STEPS_TIMED)
    eng.set_timing(False)
    P_nom = (eng.key_bits + 7) // 8
    P_exe = eng.sort_passes(sp.st.skip_low)
    per_step = {k: v["avg_ms"] * v["launches_per_step"] for k, v in ktab.items() if not k.startswith("rows_gather")}
    dom = max(per_step, key=per_step.get)
    # HBM bytes per launch of that kernel from the PMC passes of tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE in runs of
    # their own, gfx950 corrections applied there); only a profile of this very workload counts
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    kernels_of = {"reduce_hashed+giant_groups": ["reduce_hashed_kernel", "span_scan_kernel", "giant_groups_kernel"], "probe_cells": ["probe_cells_lds_kernel", "probe_cells_filtered_kernel", "probe_cells_kernel"],
                  "filter_pack": ["filter_pack_stream_kernel", "filter_pack_kernel"], "tile_count": ["tile_count_kernel"],
                  "scatter": ["scatter_kernel"], "reduce_windows+span_scan": ["reduce_hashed_kernel", "reduce_windows_kernel", "span_scan_kernel", "giant_groups_kernel"]}
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload_records") == N_total and G == 1:
                names = kernels_of.get(dom.split(" ")[0], [])
                got = [tj["kernels"][k]["hbm_bytes_per_launch"] for k in names if k in tj.get("kernels", {})]
                if got:
                    traffic = sum(got) if dom.startswith("reduce_") else got[0]
        except Exception:
            traffic = None

    out = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        d = ktab[dom]
        # algorithmic bytes of one step of the whole job: B = 24N + 4H + 8K(3+2P) + 12Z — with the draw term at what
        # the step really reads: one decision bit per hit (H / 8 bytes) unless --draw-words hands it the 32-bit draws
        # (round 4 credited 4H = 0.36 GB of draws that no kernel read any more; B_8d keeps SURVEY's letter for comparison)
        draw_b = (hits + 7) // 8 if not args.draw_words else 4 * hits

        def B_of(P, draws=None):
            return 24 * N_total + (draw_b if draws is None else draws) + 8 * K_job * (3 + 2 * P) + 12 * Z_job

        def Bread_of(P, draws=None):
            return 24 * N_total + (draw_b if draws is None else draws) + 8 * K_job * (2 + P)
        gbs = lambda b: b / (ms_step * 1e-3) / 1e9 / G      # per GPU
        out = {
            "metric": "BAM records/sec -> deduped UMI matrix; achieved HBM GB/s vs roofline",
            "value": N_total * args.steps / dt, "unit": "records/s",
            "n_gpus": G, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if G > 1 else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: %d synthetic records, 10000 barcodes x 30000 genes, --cell 1.0 --depth 1.0 --seed 926, "
                                    "uniform cells/genes, 10-bp UMIs" % N_total) if c2 else workload.describe(N_total), "scope": "device kernels, inputs resident in HBM (%s + the draw stream%s); the step ends at K3's segmented rows (rows_gather, which concatenates them where they are wanted — in the product it IS the device-to-host copy — is timed on its own line)" % ("cb array + blocked gx|umi|meta runs, the engine's staging layout" if blk is not None else "packed SoA",
                                                                                           " as 32-bit draws, turned into decisions every step" if args.draw_words else " as K1b reads it: one keep/drop decision bit per CB hit, what mt_fill_kernel + draw_bits_kernel leave in the ring in the product; every byte count here takes the draws at H/8 bytes, what is read (SURVEY 8d's 4 bytes per hit: bytes_8d / B_8d)"),
                       "record_layout": "blocked" if blk is not None else "soa",
                       "records_per_gpu": n_local, "key_bits": eng.key_bits, "radix_passes_nominal": P_nom,
                       "radix_passes_executed": P_exe,
                       "sharding": ("cell-hash, one all-to-all, %s" % ("3-stream pipeline" if sp.pipelined else "single stream")) if G > 1 else "single GPU",
                       "lookup_tables": eng.table_modes},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": d["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": d["frac"], "traffic": traffic, "bytes_per_launch": d["bytes_per_launch"], "bytes_8d": d.get("bytes_8d"),
                         "avg_launch_ms": d["avg_ms"], "launches_timed": d["launches_timed"],
                         "share_of_step": per_step[dom] / sum(per_step.values())},
            "kernels": ktab,
            "whole_path": {"per_gpu": True,
                           "B_nominal_P%d" % P_nom: B_of(P_nom), "B_executed_P%d" % P_exe: B_of(P_exe),
                           "B_read_nominal": Bread_of(P_nom), "B_read_executed": Bread_of(P_exe),
                           "GBs_nominal": gbs(B_of(P_nom)), "GBs_executed": gbs(B_of(P_exe)),
                           "frac_of_peak_nominal": gbs(B_of(P_nom)) / HBM_PEAK_GBS,
                           "frac_of_peak": gbs(B_of(P_exe)) / HBM_PEAK_GBS,
                           "read_GBs_executed": gbs(Bread_of(P_exe)), "read_frac_of_peak": gbs(Bread_of(P_exe)) / HBM_PEAK_GBS,
                           "read_frac_of_peak_nominal": gbs(Bread_of(P_nom)) / HBM_PEAK_GBS,
                           "draw_bytes_counted": draw_b,
                           "B_8d_executed": B_of(P_exe, 4 * hits), "B_read_8d_executed": Bread_of(P_exe, 4 * hits)},
            "step_with_draw_generation": with_draws,
            "counters": {"total": N_total, "hits": hits, "sampled": sampled, "valid": valid,
                         "keys": K_job, "rows": Z_job, "device_error_bits": err},
        }
        if N_total == 200_000_000 and not c2:  # the job is the same for every rank count: so are its totals
            got = {k: out["counters"][k] for k in workload.EXPECTED_200M}
            out["counters"]["same_as_single_gpu_reference_run"] = got == workload.EXPECTED_200M

    # free the resident job before the host-side legs
    del sp, cb, gx, umi, meta, d_draws, d_stream, blk
    eng.close()
    torch.cuda.empty_cache()

    if rank == 0 and G == 1 and c2:
        out["cpu_baseline"] = None             # the legs below are defined on the headline workload
    elif rank == 0 and G == 1:
        if not args.no_devpath:
            out["device_path"] = device_path_leg(job, dev, local, N_total, out["counters"])
        if not args.no_cpu:
            out["cpu_baseline"], out["parity_vs_cpu"] = cpu_leg(job, dev, local, max(1, min(args.cpu_sample // job.seg_len, workload.SEGMENTS)))
        else:
            out["cpu_baseline"] = None
        if not args.no_e2e:
            out["e2e"] = e2e_leg(job, [int(x) for x in args.e2e_records.split(",")])
        # the three scopes side by side: kernels on resident inputs (= value), host SoA -> COO on the host, BAM -> .gz
        out["scopes"] = {"kernels_only_records_per_s": out["value"],
                         "device_path_records_per_s": (out.get("device_path") or {}).get("value"),
                         "e2e_records_per_s_to_process_exit": {k: v.get("value") for k, v in (out.get("e2e") or {}).items() if isinstance(v, dict)}}
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def device_path_leg(job, dev, local, N_total, want):
    """SURVEY 8d 'device-path': packed SoA batches in pinned host memory -> hipMemcpyAsync -> kernels -> COO on the host.
    The clock starts at the first push and stops when finish() has returned the COO arrays."""
    lists = job.lists
    pb = F.PinnedBatch(N_total)
    off = 0
    for s in range(workload.SEGMENTS):
        c, g, u, m = job.segment_packed(s, dev)
        pb.fill(off, c, g, u, m)
        off += job.seg_len
    del c, g, u, m
    job._pool = None
    torch.cuda.empty_cache()
    eng = F.Engine.from_lists(lists, rate_depth=workload.RATE_DEPTH, seed=workload.SEED, umi_max_bases=workload.UMI_LEN,
                              device=local, batch_records=8 << 20, key_capacity=N_total // 4)
    runs = []
    res = None
    try:
        for rep in range(3):
            eng.reset(); eng.reseed(workload.SEED, lists.mt_skip)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.push_pinned(pb)
            t1 = time.perf_counter()
            res = eng.finish(copy=False)            # the C ABI's own arrays (the ctypes wrapper's copy of 12 bytes per row is not the path)
            t2 = time.perf_counter()
            runs.append((t2 - t0, t1 - t0, t2 - t1))
    finally:
        eng.close()
        pb.close()
    same = (res["total"], res["sampled"], res["valid"], res["nnz"]) == (want["total"], want["sampled"], want["valid"], want["rows"])
    best = min(runs)
    # the depth draws are generated on the device (mt_fill_kernel) unless FASTF_HOST_DRAWS=1 sends them over PCIe, 4 bytes per hit
    host_draws = os.environ.get("FASTF_HOST_DRAWS", "")[:1] == "1"
    bytes_h2d = 24 * N_total + (4 * want["hits"] if host_draws else 0)
    return {"value": N_total / best[0], "unit": "records/s", "seconds": best[0], "push_s": best[1], "finish_s": best[2],
            "runs_s": [r[0] for r in runs], "h2d_bytes": bytes_h2d, "h2d_GBs": bytes_h2d / best[1] / 1e9,
            "frac_of_pcie": (bytes_h2d / best[0] / 1e9) / PCIE_GBS, "pcie_peak_GBs_assumed": PCIE_GBS,
            "scope": "pinned host SoA (8 M-record chunks) -> hipMemcpyAsync on the copy stream -> K1 per chunk -> sort + reduce -> COO D2H",
            "draws": "host MT19937, 4 bytes per hit over PCIe" if host_draws else "device MT19937 (mt_fill_kernel on its own stream)",
            "same_result_as_resident_steps": bool(same)}


def cpu_leg(job, dev, local, n_seg):
    """the CPU oracle (port of the reference algorithm: hash probe + MT draw + sort/aggregate, no SQLite, no file I/O) on
    the first n_seg segments of the job, 1 core; the GPU result on that sample must equal it bit for bit"""
    from oracle import oracle as O
    lists = job.lists
    parts = [job.segment_strings(s, dev, job.seg_len) for s in range(n_seg)]
    fl, xf = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
    w = [max(p[i].dtype.itemsize for p in parts) for i in (2, 3, 4)]
    cbs, gxs, ubs = (np.concatenate([p[i].astype("S%d" % w[i - 2]) for p in parts]) for i in (2, 3, 4))
    del parts
    S = len(fl)
    t1 = time.perf_counter()
    ora = O.run_bam2db(job.bt, job.ft, fl, xf, cbs, gxs, ubs, workload.RATE_CELL, workload.RATE_DEPTH, workload.SEED)
    cpu_dt = time.perf_counter() - t1
    base = {"value": S / cpu_dt, "unit": "records/s", "cores": 1, "kind": "port",
            "sample": "first %d records (%d of %d segments) of the same workload through oracle/fastf_oracle.c, %.1f s" % (S, n_seg, workload.SEGMENTS, cpu_dt),
            "note": "the reference's own bam2db (htslib + SQLite INSERT + GROUP BY + gz writers) ran at 0.24 M records/s on 1 core "
                    "in the survey session (BASELINE.md section 2); it cannot be built here (htslib absent)"}
    # the product's own packer on the strings, against the generator's direct packing, then the device result
    cbk, gxk, umi, meta = F.pack_records(lists, fl, xf, cbs, gxs, ubs)
    pack_ok = True
    for s_ in range(n_seg):
        a, b = s_ * job.seg_len, (s_ + 1) * job.seg_len
        c, g, u, m = (hostmem.to_host(t) for t in job.segment_packed(s_, dev))
        nn = (meta[a:b] & 4) != 0
        pack_ok = pack_ok and (np.array_equal(cbk[a:b].view(np.int64), c) and np.array_equal(gxk[a:b].view(np.int64), g)
                               and np.array_equal(meta[a:b].view(np.int32), m) and np.array_equal(umi[a:b].view(np.int32)[nn], u[nn]))
    e2 = F.Engine.from_lists(lists, rate_depth=workload.RATE_DEPTH, seed=workload.SEED, umi_max_bases=workload.UMI_LEN, device=local)
    try:
        e2.push(cbk, gxk, umi, meta)
        res = e2.finish()
    finally:
        e2.close()
    ok = pack_ok and (res["total"], res["sampled"], res["valid"], res["nnz"]) == (ora["total"], ora["sampled"], ora["valid"], ora["nnz"]) \
        and np.array_equal(res["cell"], ora["cell"].astype(np.uint32)) \
        and np.array_equal(res["feature"], ora["feature"].astype(np.uint32)) \
        and np.array_equal(res["count"], ora["count"].astype(np.uint32))
    return base, ("bit-exact" if ok else "MISMATCH")


def e2e_leg(job, sizes):
    """the real CLI (fastF bam2db -c .5 -r .5) on generated BAM files, timed from process start to process EXIT (the time
    to "the three .gz files closed" beside it)"""
    threads = int(os.environ.get("FASTF_HOST_THREADS", "16"))
    gen = os.path.join(ROOT, "build", "gen_bam")
    os.makedirs(os.path.dirname(gen), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-o", gen, os.path.join(ROOT, "tools", "gen_bam.c"), "-lz", "-lpthread"])
    cli = os.path.join(ROOT, "fastf_amd", "bin", "fastF")
    out = {"host_threads": threads, "flags": "-c 0.5 -r 0.5 -s 926", "lists": "%d barcodes x %d genes" % (workload.N_BARCODES, workload.N_GENES)}
    tmp_root = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=tmp_root) as td:
        open(os.path.join(td, "bar.tsv"), "wb").write(job.bt)
        open(os.path.join(td, "feat.tsv"), "wb").write(job.ft)
        # the third file is the configs[2]-sized one: the Cell-Ranger-shaped body written `rep` times behind one header
        # (every record ten times: ten times the keys, the same distinct UMIs — the same matrix as the 20 M-record file)
        big_rep = max(1, int(os.environ.get("FASTF_E2E_BIG_REPEAT", "10")))
        cases = [("skinny", sizes[0], 0, 1, 1), ("cell_ranger_shaped", sizes[1], 91, 1, 1)]
        if big_rep > 1 and not os.environ.get("FASTF_E2E_NO_BIG"):
            # the configs[2]-sized file twice: ten DIFFERENT bodies from ten seeds (200 M distinct records: the matrix, the depth of
            # its groups and the output files are those of a file of that size) — the line of record — and, for continuity with
            # round 4, the one body written ten times (the matrix of the 20 M-record file, every group ten times deeper)
            cases.append(("cell_ranger_shaped_%dM" % (sizes[1] * big_rep // 1_000_000), sizes[1], 91, 1, big_rep))
            if not os.environ.get("FASTF_E2E_NO_REPEATED"):
                cases.append(("cell_ranger_shaped_%dM_repeated_body" % (sizes[1] * big_rep // 1_000_000), sizes[1], 91, big_rep, 1))
        for label, n_gen, seq_len, rep, bodies in cases:
            n = n_gen * rep * bodies
            bam = os.path.join(td, "in.bam")
            t0 = time.perf_counter()
            subprocess.check_call([gen, bam, os.path.join(td, "bar.tsv"), os.path.join(td, "feat.tsv"), str(n_gen), "7", "12", str(seq_len), str(threads), str(rep), str(bodies)])
            t_gen = time.perf_counter() - t0
            out[label] = {"records": n, "bam_bytes": os.path.getsize(bam), "bam_generated_in_s": t_gen, "records_distinct": rep == 1}
            if rep > 1:
                out[label]["built_as"] = "the %d-record body written %d times behind one header" % (n_gen, rep)
            if bodies > 1:
                out[label]["built_as"] = "%d bodies of %d records from %d seeds behind one header" % (bodies, n_gen, bodies)
            # the same file with the BGZF inflate on the host's threads only, and shared with the device (the CLI's default)
            for variant, extra in (("host_inflate", {"FASTF_GPU_INFLATE": "0"}), ("hybrid_inflate", {"FASTF_GPU_INFLATE": "1"})):
                best = None
                for _rep in range(2 if rep * bodies == 1 else 1):
                    od = os.path.join(td, "out"); os.makedirs(od, exist_ok=True)
                    for f in os.listdir(od):
                        os.unlink(os.path.join(od, f))
                    env = dict(os.environ, FASTF_HOST_THREADS=str(threads), FASTF_PROFILE="1", FASTF_BAM_PROFILE="1", **extra)
                    t0 = time.perf_counter(); w0 = time.time()
                    p = subprocess.run([cli, "bam2db", "-b", bam, "-a", os.path.join(td, "bar.tsv"), "-f", os.path.join(td, "feat.tsv"),
                                        "-o", od, "-c", "0.5", "-r", "0.5", "-s", "926"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                    wall = time.perf_counter() - t0
                    if p.returncode != 0:
                        best = {"error": p.stderr.decode(errors="replace")[-400:]}
                        break
                    lines = p.stderr.decode(errors="replace").splitlines()
                    prof = [l for l in lines if l.startswith("[bam2db] lists")]
                    rdr = [l for l in lines if l.startswith("[bam] ")]
                    closed = [float(l.split(" at ")[1].split()[0]) for l in lines if l.startswith("[bam2db] outputs closed at")]
                    # end-to-end = process start -> the .gz files closed; what follows (the kernel unmapping the
                    # BAM, the pinned slab and the GPU context of the exiting process) is reported next to it, not in it
                    done = (closed[-1] - w0) if closed else wall
                    md5 = subprocess.run("zcat %s/matrix.mtx.gz | md5sum" % od, shell=True, stdout=subprocess.PIPE).stdout.decode().split()[0]
                    # (the header names the BAM's path: the rows alone compare runs on copies of the file)
                    body_md5 = subprocess.run("zcat %s/matrix.mtx.gz | grep -v '^%%' | md5sum" % od, shell=True, stdout=subprocess.PIPE).stdout.decode().split()[0]
                    dims = subprocess.run("zcat %s/matrix.mtx.gz | grep -v '^%%' -m 1" % od, shell=True, stdout=subprocess.PIPE).stdout.decode().split()
                    # the clock a user lives with: process start -> process exit.  "outputs closed" (SURVEY 8d's end-to-end
                    # scope) is kept beside it
                    # the three pieces of the run (bam2db()'s own clock, FASTF_PROFILE): until the device takes work, the steady
                    # state (records decoded from then on: reader windows shared with the device), the way out
                    ph = {}
                    for l in lines:
                        if l.startswith("[bam2db] phases:"):
                            ph = {k: float(v) for k, v in (kv.split("=") for kv in l.split(":", 1)[1].split())}
                    if best is None or wall < best["seconds"]:
                        best = {"value": n / wall, "unit": "records/s", "seconds": wall, "scope": "process start -> process exit",
                                "seconds_to_outputs_closed": done, "records_per_s_to_outputs_closed": n / done, "matrix_md5": md5, "matrix_rows_md5": body_md5,
                                "matrix_rows": int(dims[2]) if len(dims) == 3 else None,
                                "start_up_s": ph.get("decoder_saw_engine_s"), "records_decoded_during_start_up": ph.get("records_before"),
                                "steady_state_records_per_s": (ph["steady_records"] / ph["steady_s"]) if ph.get("steady_s") else None,
                                "steady_state_s": ph.get("steady_s"), "finish_and_write_s": (ph["outputs_closed_s"] - ph["last_record_s"]) if ph else None,
                                "exit_s": wall - done,
                                "stages": prof[-1] if prof else "", "reader": " | ".join(rdr)}
                out[label][variant] = best
            v = [out[label][k] for k in ("host_inflate", "hybrid_inflate") if "value" in out[label][k]]
            if v:
                out[label]["value"] = max(x["value"] for x in v); out[label]["unit"] = "records/s"
                out[label]["same_matrix"] = len({x["matrix_md5"] for x in v}) == 1
            if bodies > 1 and v and not os.environ.get("FASTF_E2E_NO_COLD"):
                # every line above reads its BAM from /dev/shm (page cache: "read 0.000 s").  ONE line with the read in it: the
                # same file on a disk-backed directory, its pages dropped from the page cache before the run
                try:
                    out[label + "_cold"] = _cold_run(cli, bam, td, n, threads, out[label]["hybrid_inflate"].get("matrix_rows_md5"))
                except Exception as e:                           # (a full disk, a filesystem without fadvise: the line is an extra, the bench goes on)
                    out[label + "_cold"] = {"skipped": "%s: %s" % (type(e).__name__, e)}
            os.unlink(bam)
    return out


def _disk_dir(need_bytes):
    """a directory on a disk-backed filesystem with room for need_bytes (FASTF_E2E_DISK_DIR first), or None"""
    fstype = {}
    try:
        for ln in open("/proc/mounts"):
            f = ln.split()
            fstype[f[1]] = f[2]
    except OSError:
        pass

    def fs_of(path):
        path = os.path.realpath(path)
        best = ""
        for m in fstype:
            if (path == m or path.startswith(m.rstrip("/") + "/")) and len(m) > len(best):
                best = m
        return fstype.get(best, "?")
    for d in [os.environ.get("FASTF_E2E_DISK_DIR"), os.path.join(ROOT, "build"), "/var/tmp", "/tmp"]:
        if not d or not os.path.isdir(d):
            continue
        if fs_of(d) in ("tmpfs", "ramfs", "devtmpfs"):
            continue
        st = os.statvfs(d)
        if st.f_bavail * st.f_frsize >= need_bytes + (2 << 30):
            return d, fs_of(d)
    return None, None


def _cold_run(cli, bam, td, n, threads, want_md5):
    import shutil
    size = os.path.getsize(bam)
    d, fs = _disk_dir(size)
    if d is None:
        return {"skipped": "no disk-backed directory with %.1f GB free (set FASTF_E2E_DISK_DIR)" % (size / 1e9)}
    cold = os.path.join(d, "fastf_cold_%d.bam" % os.getpid())
    try:
        t0 = time.perf_counter()
        shutil.copyfile(bam, cold)
        fd = os.open(cold, os.O_RDONLY)
        try:
            os.fsync(fd)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)           # clean pages of the file leave the page cache
        finally:
            os.close(fd)
        t_copy = time.perf_counter() - t0
        resident = None
        try:                                                             # how much of it is still cached (util-linux fincore, if there)
            r = subprocess.run(["fincore", "--bytes", "--noheadings", "--output", "RES", cold], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            if r.returncode == 0:
                resident = int(r.stdout.decode().split()[0]) / size
        except (OSError, ValueError, IndexError):
            pass
        od = os.path.join(td, "out_cold"); os.makedirs(od, exist_ok=True)
        env = dict(os.environ, FASTF_HOST_THREADS=str(threads), FASTF_PROFILE="1", FASTF_BAM_PROFILE="1", FASTF_GPU_INFLATE="1")
        t0 = time.perf_counter(); w0 = time.time()
        p = subprocess.run([cli, "bam2db", "-b", cold, "-a", os.path.join(td, "bar.tsv"), "-f", os.path.join(td, "feat.tsv"),
                            "-o", od, "-c", "0.5", "-r", "0.5", "-s", "926"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": p.stderr.decode(errors="replace")[-400:]}
        lines = p.stderr.decode(errors="replace").splitlines()
        closed = [float(l.split(" at ")[1].split()[0]) for l in lines if l.startswith("[bam2db] outputs closed at")]
        done = (closed[-1] - w0) if closed else wall
        md5 = subprocess.run("zcat %s/matrix.mtx.gz | grep -v '^%%' | md5sum" % od, shell=True, stdout=subprocess.PIPE).stdout.decode().split()[0]
        rdr = [l for l in lines if l.startswith("[bam] ") and "read " in l]
        return {"value": n / wall, "unit": "records/s", "seconds": wall, "scope": "process start -> process exit, the BAM on a disk-backed "
                "filesystem with its pages dropped from the page cache before the run (posix_fadvise DONTNEED after fsync)",
                "seconds_to_outputs_closed": done, "records_per_s_to_outputs_closed": n / done, "bam_bytes": size,
                "bam_bytes_per_s": size / wall, "filesystem": fs, "directory": d, "copied_and_dropped_in_s": t_copy,
                "cached_fraction_before_the_run": resident, "matrix_rows_md5": md5, "same_matrix_rows_as_the_cached_run": md5 == want_md5,
                "reader": " | ".join(rdr[:1])}
    finally:
        try:
            os.unlink(cold)
        except OSError:
            pass


if __name__ == "__main__":
    main()
