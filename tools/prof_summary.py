#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into the
summaries committed under profiles/.  Usage:
   tools/prof_summary.py <round-tag> <stats_dir> <fetch_dir> <write_dir> [n_keys]
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
under-reports wide coalesced streaming reads by exactly 2x (re-checked here on tile_count_kernel,
whose algorithmic read is known: 8 B per key); WRITE_SIZE is exact for streaming stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("fastf::", "")
    for k in ("scatter_kernel", "filter_pack_stream_kernel", "filter_pack_kernel", "probe_cells_lds_kernel", "reduce_windows_kernel",
              "reduce_hashed_kernel", "rows_gather_kernel"):
        if n.startswith(k + "<"):
            return k                                                          # template instantiations of one kernel
    return n


def counters(d, cname):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    tag, sdir, fdir, wdir = sys.argv[1:5]
    n_keys = int(sys.argv[5]) if len(sys.argv) > 5 else 10_000_000
    n_records = int(sys.argv[6]) if len(sys.argv) > 6 else None
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats = glob.glob(os.path.join(sdir, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    agg = {}
    for r in rows:                      # the 8 digit-shift instantiations of scatter_kernel are one kernel
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0, 0.0, 1e30, 0.0])
        a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"]); a[2] += float(r["Percentage"])
        a[3] = min(a[3], float(r["MinNs"])); a[4] = max(a[4], float(r["MaxNs"]))
    with open(os.path.join(out_dir, "%s_kernel_stats.csv" % tag), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, a[0], "%.0f" % a[1], "%.0f" % (a[1] / a[0]), "%.2f" % a[2], "%.0f" % a[3], "%.0f" % a[4]])
    fetch, write = counters(fdir, "FETCH_SIZE"), counters(wdir, "WRITE_SIZE")
    tot, calls = defaultdict(float), defaultdict(int)
    for r in rows:
        tot[short(r["Name"])] += float(r["TotalDurationNs"]); calls[short(r["Name"])] += int(r["Calls"])
    avg = {k: tot[k] / calls[k] for k in tot}
    tc = fetch.get("tile_count_kernel<false>") or fetch.get("tile_count_kernel") or []   # <false> = the contiguous passes
    cal = (8.0 * n_keys) / (sum(tc) / len(tc) * 1024) if tc else None
    summ = {"_note": "per-launch averages; FETCH/WRITE in KiB as reported by rocprofv3 --pmc (separate passes); "
                     "hbm_bytes = FETCH*1024*2 (gfx950 correction) + WRITE*1024",
            "fetch_factor_observed_on_tile_count_kernel": cal, "n_keys": n_keys, "workload_records": n_records, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith("__amd") or k.startswith("at::") or "at::native" in k or "void at::" in k:
            continue
        fl, wl = fetch.get(k, []), write.get(k, [])
        f = sum(fl) / len(fl) if fl else 0.0
        wv = sum(wl) / len(wl) if wl else 0.0
        summ["kernels"][k] = {"launches_sampled": len(fl), "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": wv,
                              "hbm_bytes_per_launch": f * 1024 * 2.0 + wv * 1024, "avg_ns_unprofiled_stats_run": avg.get(k)}
    sc = summ["kernels"].get("scatter_kernel")
    if sc:
        summ["scatter_kernel_hbm_bytes_per_launch"] = sc["hbm_bytes_per_launch"]
        summ["scatter_kernel_algorithmic_bytes_per_launch"] = 16 * n_keys
    json.dump(summ, open(os.path.join(out_dir, "%s_traffic.json" % tag), "w"), indent=1)
    json.dump(summ, open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
