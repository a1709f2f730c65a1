#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into
profiles/pmc_traffic.json:  HBM-side bytes per launch of one kernel.

    tools/pmc_summarize.py <fetch_dir> <write_dir> <kernel-name substring> <json key> [n_envs] [algorithmic bytes per env-step]

Corrections per MI355X_MICROARCH.md §HBM and the calibration recorded in the same JSON: unit KiB,
FETCH_SIZE x2 on gfx950, WRITE_SIZE x1."""
import csv
import glob
import json
import os
import sys


def avg(d, counter, kern):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter and kern in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for a kernel matching {kern!r} under {d}")
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_dir, write_dir, kern, key = sys.argv[1:5]
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 20
    alg = int(sys.argv[6]) if len(sys.argv) > 6 else 120
    f_kb, nf = avg(fetch_dir, "FETCH_SIZE", kern)
    w_kb, nw = avg(write_dir, "WRITE_SIZE", kern)
    fetch = 2.0 * f_kb * 1024.0
    write = w_kb * 1024.0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "pmc_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[key] = {
        "kernel_match": kern, "dispatches": min(nf, nw), "FETCH_SIZE_KB_avg_raw": f_kb, "WRITE_SIZE_KB_avg": w_kb,
        "fetch_bytes_corrected": fetch, "write_bytes": write, "bytes_per_launch": fetch + write,
        "bytes_per_env_step": (fetch + write) / n, "algorithmic_bytes_per_env_step": alg, "ratio": (fetch + write) / n / alg, "n_envs": n,
    }
    if "cartpole_specialised_2p20" in key:
        doc["step_kernel_cartpole_bytes_per_launch"] = fetch + write   # what bench.py reports as roofline.traffic
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc[key], indent=1))


if __name__ == "__main__":
    main()
