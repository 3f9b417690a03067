#!/usr/bin/env python3
"""Fold two rocprofv3 `--pmc` passes (FETCH_SIZE, WRITE_SIZE) into profiles/<round>_hbm_traffic.json, PER DISPATCH.

usage: tools/hbm_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv OUT.json

rocprofv3 reports both counters in KB per dispatch.  One outer iteration of the benchmark launches
    oc_solve (lean kernel, full batch)  ->  oc_solve (exact-capable kernel, resumes what the lean one left: normally
    nothing, every workgroup returns at once)  ->  aux_riccati  ->  aux_forward,
and bench.py adds a 4-trajectory parity solve before the loop.  Only FULL-BATCH dispatches describe the timed launch:
they are selected by the largest Grid_Size of each kernel, the lean OC kernel apart from the resume kernel by its
template arguments, and every one is listed (`dispatches`) beside the mean the bench line quotes (`hbm_bytes_per_launch`).
The k-th selected dispatch of the FETCH pass is paired with the k-th of the WRITE pass (same command, same order).

Calibration (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads (x2 there);
for the 4-8-byte-per-lane accesses of these kernels it is uncalibrated, so the raw figure is a LOWER bound of the bytes
read; WRITE_SIZE is exact for streaming stores.  Both count Infinity-Cache hits: this is memory-side traffic, not DRAM.
"""
import csv
import os
import json
import re
import sys
from collections import defaultdict

KERNELS = {"oc_solve": "oc_solve_kernel", "oc_solve_resume": "oc_solve_kernel", "oc_solve_wide": "oc_solve_wide_kernel", "oc_solve_wide_w4": "oc_solve_wide_kernel",
           "aux_riccati": "aux_riccati_kernel", "aux_forward": "aux_forward_kernel"}


def classify(name):
    """bench key of a kernel name, or None."""
    if "oc_solve_wide_kernel" in name:
        # oc_solve_wide_kernel<Model, T, EXACT, BND, W>: W > 1 is the launch with several wavefronts per trajectory (round 6: the second
        # launch of a two-launch solve, or the whole solve of a small batch)
        m = re.search(r"oc_solve_wide_kernel<.*,\s*(\d+)\s*>\s*\(", name)
        return "oc_solve_wide_w4" if (m and int(m.group(1)) > 1) else "oc_solve_wide"
    if "oc_solve_kernel" in name:
        # oc_solve_kernel<Model, T, G, EXACT, PK>: EXACT = true is the resume launch of the two-launch solve
        m = re.search(r"oc_solve_kernel<.*?,\s*(float|double),\s*\d+,\s*(true|false)", name)
        if m and m.group(1) == "float" and os.environ.get("LFSD_PROFILE_DTYPE") == "f64":
            # the fp32 solve that seeds a cold fp64 solve (lfsd_capi.cpp, coc_solve_seeded), and its (empty) resume launch
            return "oc_solve_seed_f32_resume" if m.group(2) == "true" else "oc_solve_seed_f32"
        return "oc_solve_resume" if (m and m.group(2) == "true") else "oc_solve"
    if "aux_riccati_kernel" in name:
        return "aux_riccati"
    if "aux_forward_kernel" in name:
        return "aux_forward"
    return None


def dispatches(path, counter):
    """{key: [(dispatch_id, grid_size, KB)]} in dispatch order."""
    out = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            key = classify(row["Kernel_Name"])
            if key:
                out[key].append((int(row["Dispatch_Id"]), int(row["Grid_Size"]), float(row["Counter_Value"])))
    for v in out.values():
        v.sort()
    return out


def full_batch(rows):
    g = max(r[1] for r in rows)
    return [r for r in rows if r[1] == g]


def main(argv):
    fetch_csv, write_csv, out = argv[1:4]
    fd, wd = dispatches(fetch_csv, "FETCH_SIZE"), dispatches(write_csv, "WRITE_SIZE")
    res = {}
    for key in sorted(set(fd) & set(wd)):
        ff, ww = full_batch(fd[key]), full_batch(wd[key])
        n = min(len(ff), len(ww))
        if n == 0:
            continue
        per = [{"grid_threads": ff[i][1], "FETCH_SIZE_KB": ff[i][2], "WRITE_SIZE_KB": ww[i][2],
                "bytes": (ff[i][2] + ww[i][2]) * 1024.0} for i in range(n)]
        mean = sum(p["bytes"] for p in per) / n
        res[key] = {"full_batch_launches": n, "dispatches_of_this_kernel_in_the_run": len(fd[key]),
                    "hbm_bytes_per_launch": mean,
                    "FETCH_SIZE_KB_per_launch": sum(p["FETCH_SIZE_KB"] for p in per) / n,
                    "WRITE_SIZE_KB_per_launch": sum(p["WRITE_SIZE_KB"] for p in per) / n,
                    "dispatches": per}
    res["_note"] = ("raw rocprofv3 FETCH_SIZE / WRITE_SIZE (KB) of separate --pmc passes, per full-batch dispatch; "
                    "FETCH_SIZE is uncalibrated (a lower bound) for 4-8-byte-per-lane accesses and both include "
                    "Infinity-Cache hits (MI355X_MICROARCH.md, HBM)")
    import os
    res["collected_at_commit"] = os.environ.get("LFSD_COMMIT")
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    res.pop("collected_at_commit")
    print(json.dumps({k: {"launches": v["full_batch_launches"], "MB_per_launch": round(v["hbm_bytes_per_launch"] / 1e6, 1),
                          "each_MB": [round(p["bytes"] / 1e6, 1) for p in v["dispatches"]]}
                      for k, v in res.items() if not k.startswith("_")}))


if __name__ == "__main__":
    main(sys.argv)
