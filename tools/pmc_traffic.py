#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of three runs of `python3 bench.py` (kernel trace + stats, --pmc
FETCH_SIZE, --pmc WRITE_SIZE; see tools/profile_round.sh) into the summaries committed under
profiles/: per-kernel average duration of the batch launches and HBM bytes per launch.

Counter handling follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so
the fetch side is reported both raw and doubled (our loads are 4 B/lane, an access width the
guide calls uncalibrated: the true value lies between the two; `hbm_bytes_per_launch` uses the
doubled, i.e. pessimistic, figure).
"""
import collections
import csv
import glob
import json
import os
import sys

# first match wins: k_refine_int must be tested before k_refine
SLOT = [("k_prologue", "prologue"), ("k_match_fast", "match_fast"), ("k_match_exact", "match_exact"),
        ("k_refine_int", "refine_int"), ("k_refine", "refine"), ("k_fill", "fill")]


def slot(name):
    for k, v in SLOT:
        if k in name:
            return v
    return None


def main(prof_dir, out_dir, tag, pairs):
    pairs = int(pairs)
    res = {"pairs_per_launch": pairs, "source": f"rocprofv3 runs of `python3 bench.py` ({tag})", "kernels": {}}
    # --- kernel trace: average duration of the batch launches (largest grid per kernel)
    trace = glob.glob(os.path.join(prof_dir, "trace", "*", "*_kernel_trace.csv"))[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        s = slot(r["Kernel_Name"])
        if s:
            dur[s].append((int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]),
                           int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for s, v in dur.items():
        gmax = max(g for g, _ in v)
        d = [t for g, t in v if g == gmax]
        res["kernels"][s] = {"launches": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d)}
    # --- PMC passes
    for cname, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        f = glob.glob(os.path.join(prof_dir, sub, "*", "*_counter_collection.csv"))[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            s = slot(r["Kernel_Name"])
            if s and r["Counter_Name"] == cname:
                acc[s].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
        for s, v in acc.items():
            gmax = max(g for g, _ in v)
            vals = [c for g, c in v if g == gmax]
            res["kernels"].setdefault(s, {})[cname + "_KiB"] = sum(vals) / len(vals)
    for s, k in res["kernels"].items():
        if "FETCH_SIZE_KiB" in k and "WRITE_SIZE_KiB" in k:
            k["fetch_bytes_raw"] = k["FETCH_SIZE_KiB"] * 1024
            k["fetch_bytes_doubled"] = 2 * k["FETCH_SIZE_KiB"] * 1024
            k["write_bytes"] = k["WRITE_SIZE_KiB"] * 1024
            k["hbm_bytes_per_launch"] = k["fetch_bytes_doubled"] + k["write_bytes"]
    os.makedirs(out_dir, exist_ok=True)
    json.dump(res, open(os.path.join(out_dir, f"{tag}_traffic.json"), "w"), indent=1)
    stats = glob.glob(os.path.join(prof_dir, "trace", "*", "*_kernel_stats.csv"))[0]
    open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w").write(open(stats).read())
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
