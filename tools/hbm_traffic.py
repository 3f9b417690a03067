#!/usr/bin/env python3
"""Fold two rocprofv3 `--pmc` passes (FETCH_SIZE, WRITE_SIZE) into profiles/<round>_hbm_traffic.json.

usage: tools/hbm_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv OUT.json [launches_per_step...]

The counters are reported by rocprofv3 in KB per dispatch. bench.py reads `hbm_bytes_per_step` of the dominant kernel
for the `roofline.traffic` field. `oc_solve` is launched twice per outer iteration (lean kernel + Newton-capable
kernel), the two auxiliary sweeps once.
"""
import csv
import json
import sys
from collections import defaultdict

KERNELS = {"oc_solve": "oc_solve_kernel", "aux_riccati": "aux_riccati_kernel", "aux_forward": "aux_forward_kernel"}
LAUNCHES_PER_STEP = {"oc_solve": 2, "aux_riccati": 1, "aux_forward": 1}
NOTE = ("raw rocprofv3 counters (KB) from separate --pmc passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline` "
        "(3 steps); oc_solve is two launches per step (lean + Newton-capable kernel); accesses are 4-B-per-lane "
        "scalar loads/stores for which MI355X_MICROARCH.md gives no FETCH_SIZE calibration (its x2 correction is for "
        "16-B-per-lane streams)")


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            for key, pat in KERNELS.items():
                if pat in row["Kernel_Name"]:
                    tot[key] += float(row["Counter_Value"])
                    cnt[key] += 1
    return tot, cnt


def main(argv):
    fetch_csv, write_csv, out = argv[1:4]
    ft, fc = per_kernel(fetch_csv, "FETCH_SIZE")
    wt, wc = per_kernel(write_csv, "WRITE_SIZE")
    res = {}
    for key in KERNELS:
        if not fc[key] or not wc[key]:
            continue
        f_kb, w_kb = ft[key] / fc[key], wt[key] / wc[key]
        per_launch = (f_kb + w_kb) * 1024.0
        res[key] = {
            "FETCH_SIZE_KB_per_launch": f_kb, "launches_fetch": fc[key],
            "WRITE_SIZE_KB_per_launch": w_kb, "launches_write": wc[key],
            "hbm_bytes_per_launch": per_launch,
            "hbm_bytes_per_step": per_launch * LAUNCHES_PER_STEP[key],
            "note": NOTE,
        }
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_step"] / 1e6, 1) for k, v in res.items()}), "MB/step")


if __name__ == "__main__":
    main(sys.argv)
