#!/usr/bin/env python3
"""Summarise a rocprofv3 CSV output directory into the small text files kept under profiles/.

  prof_summary.py stats <dir>          per-kernel launches / total / avg / min / max (from *_kernel_trace.csv)
  prof_summary.py pmc   <dir>          per-kernel counter sums and the largest launch's values (from *_counter_collection.csv)
  prof_summary.py traffic <fetch_dir> <write_dir> <log_n>   HBM bytes per launch of the bench kernels as JSON

HBM bytes from the PMC counters follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a
16-B-per-lane coalesced streaming read, so reads = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.match(r"(?:void )?(?:zkk::)?([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits


def stats(d):
    per = defaultdict(list)
    for f in find(d, "_kernel_trace.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in per.values())
    print(f"{'kernel':44s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:44s} {len(v):6d} {sum(v):12.1f} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f} {100 * sum(v) / tot:6.1f}")


def pmc(d):
    rows = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [(dispatch, value, dur_us)]
    for f in find(d, "_counter_collection.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                rows[short(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), dur))
    for k, ctrs in sorted(rows.items()):
        for cname, vals in sorted(ctrs.items()):
            big = max(vals, key=lambda t: t[1])
            line = f"{k:40s} {cname:22s} launches {len(vals):4d}  sum {sum(v for _, v, _ in vals):.6g}  largest launch {big[1]:.6g} (dispatch {big[0]})"
            if cname == "FETCH_SIZE":
                line += f"  => HBM read {2 * big[1] * 1024 / 1e9:.4f} GB (2 x FETCH_SIZE KiB, gfx950 correction)"
            if cname == "WRITE_SIZE":
                line += f"  => HBM write {big[1] * 1024 / 1e9:.4f} GB"
            print(line)


def traffic_dict(rd_dir, wr_dir, log_n):
    """HBM bytes per launch of the bench kernels from a FETCH_SIZE pass and a WRITE_SIZE pass (what bench.py reports as
    roofline.traffic: from its own counter passes when rocprofv3 is there, else from the committed file of the same command)."""
    names = {"k_multifold_lazy": "multifold", "k_multifold": "multifold", "k_multifold_roll": "multifold", "k_multifold_wide": "multifold_wide", "k_multifold_wide_lds": "multifold_wide", "k_multifold_wide_dma": "multifold_wide", "k_multifold_wide_mfma": "multifold_wide", "k_block_sums_dma": "block_sums", "k_multifold_coop": "multifold_tail", "k_block_sums": "block_sums",
             "k_sum9": "sum9", "k_sum9_lazy": "sum9", "k_sum9_mfma": "sum9", "k_sum16_mfma": "sum9", "k_cross64_mfma": "sum9", "k_fold2_tables": "fold2x4_sum9", "k_fold2_tables_dma": "fold2x4_sum9",
             "k_fold3_tables_dma": "fold2x4_sum9", "k_fold3_tables": "fold2x4_sum9", "k_fold2_sum9": "fold2x4_sum9", "k_fold_sum27": "fold2x4_sum9", "k_fold3_cross64": "fold2x4_sum9",
             "k_fold3_cross64_mfma": "fold2x4_sum9", "k_eval3": "eval3", "k_fold4_eval3": "fold4_eval3"}
    acc = defaultdict(lambda: {"launches": 0, "read_bytes": 0.0, "write_bytes": 0.0})
    for d, counter, key, scale in ((rd_dir, "FETCH_SIZE", "read_bytes", 2 * 1024.0), (wr_dir, "WRITE_SIZE", "write_bytes", 1024.0)):
        for f in find(d, "_counter_collection.csv"):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    base = re.sub(r"<.*", "", short(r["Kernel_Name"]))
                    if r["Counter_Name"] != counter or base not in names:
                        continue
                    a = acc[names[base]]
                    a[key] += float(r["Counter_Value"]) * scale
                    if counter == "FETCH_SIZE":
                        a["launches"] += 1
    for a in acc.values():
        a["bytes_per_launch"] = (a["read_bytes"] + a["write_bytes"]) / max(a["launches"], 1)
    return {
        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of the bench command; tools/prof_summary.py traffic",
        "correction": "HBM read bytes = 2 * FETCH_SIZE KiB * 1024 (gfx950, 16 B/lane streams); write bytes = WRITE_SIZE KiB * 1024",
        "log_n": int(log_n), "kernels": dict(acc)}


def traffic_json(rd_dir, wr_dir, log_n):
    import json

    print(json.dumps(traffic_dict(rd_dir, wr_dir, log_n), indent=1))


if __name__ == "__main__":
    if len(sys.argv) == 5 and sys.argv[1] == "traffic":
        traffic_json(sys.argv[2], sys.argv[3], sys.argv[4])
    elif len(sys.argv) == 3 and sys.argv[1] in ("stats", "pmc"):
        (stats if sys.argv[1] == "stats" else pmc)(sys.argv[2])
    else:
        raise SystemExit(__doc__)
