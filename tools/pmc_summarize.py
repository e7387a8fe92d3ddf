#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc counter CSVs into profiles/pmc_traffic.json.

Usage (after tools/pmc_passes.sh ran on the GPU box and its CSVs were copied to profiles/<tag>_pmc/):
    python tools/pmc_summarize.py profiles/r01_pmc

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE under-reports by a factor the
calibration kernel (tools/pmc_calib.hip, a 1 GiB coalesced streaming read / write) measures; the
factor is applied to every kernel.  Separate passes per counter, as MI355X_MICROARCH.md prescribes.
"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["k_linearize_lm", "k_linearize_pose", "k_schur", "k_assemble", "k_backsub_chi2<true, true>", "k_potrf_inv", "k_trsm", "k_back_solve",
           "k_match_hamming256", "k_match_rowbucket", "k_gather_edges"]


def per_kernel(path):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: v[0] / v[1] * 1024.0 for k, v in acc.items()}  # bytes per launch


def pick(table, key):
    hits = [v for k, v in table.items() if key in k]
    return hits[0] if hits else None


def main(d):
    calib_r = per_kernel(os.path.join(d, "calib_FETCH_SIZE_counter_collection.csv"))
    calib_w = per_kernel(os.path.join(d, "calib_WRITE_SIZE_counter_collection.csv"))
    gib = float(1 << 30)
    fetch_corr = gib / pick(calib_r, "calib_read8")
    write_corr = gib / pick(calib_w, "calib_write8")
    rd = per_kernel(os.path.join(d, "bench_FETCH_SIZE_counter_collection.csv"))
    wr = per_kernel(os.path.join(d, "bench_WRITE_SIZE_counter_collection.csv"))
    out = {
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md HBM section); counters are in KiB; "
                  "both are calibrated on 1 GiB coalesced 8 B/lane streaming kernels (tools/pmc_calib.hip): FETCH_SIZE reports ~1/2 of the bytes "
                  "on gfx950, WRITE_SIZE is exact; values are bytes per launch averaged over every launch of `bench.py --steps 10 --no-replay` at config 4 "
                  "(all launches inside LM loops); these are bytes at the L2 <-> fabric boundary: Infinity-Cache hits are in them",
        "source": os.path.relpath(d, ROOT),
        "fetch_correction": fetch_corr,
        "write_correction": write_corr,
        "kernels": {},
    }
    for k in KERNELS:
        r, w = pick(rd, k), pick(wr, k)
        if r is None or w is None:
            continue
        out["kernels"][k] = {"read_bytes": r * fetch_corr, "write_bytes": w * write_corr, "hbm_bytes": r * fetch_corr + w * write_corr}
    ks = out["kernels"]
    if "k_linearize_lm" in ks and "k_linearize_pose" in ks:
        out["sweep_hbm_bytes_per_launch"] = ks["k_linearize_lm"]["hbm_bytes"] + ks["k_linearize_pose"]["hbm_bytes"]
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
