#!/usr/bin/env python3
"""Fold rocprofv3 `--pmc` SQ passes into profiles/<round>_issue_counters.json (bench.py reads `valu_issue_util` and
`mfma_busy` of the dominant kernel for its `roofline` object).

usage: tools/issue_counters.py OUT.json PASS1_counter_collection.csv [PASS2_counter_collection.csv ...]

Units (MI355X_MICROARCH.md, per-instruction cycle constants): SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles summed
over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs.
  valu_issue_util = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES      share of a resident wave's lifetime it spends issuing VALU work
  mfma_busy       = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES / waves_per_simd)   share of the SIMD-time covered by
                    waves during which the matrix pipe is busy (waves_per_simd from the kernel's register footprint)
  valu_lane_util  = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU)   mean share of the 64 lanes that are EXEC-enabled per vector
                    instruction (lanes that execute REDUNDANT group-uniform work count as enabled: an upper bound of the
                    useful share, the counter-side cross-check of perf_model's useful-flop figure)
  valu_flops_executed = 2 * FMA_F32 + ADD_F32 + MUL_F32 + TRANS_F32 instruction counts x 64 x valu_lane_util (packed
                    instructions counted as the hardware reports them): what the vector pipe executed, useful or not
Only FULL-BATCH dispatches (largest Grid_Size per kernel) of the lean OC kernel / the two sweeps are folded."""
import csv
import json
import sys
from collections import defaultdict

from hbm_traffic import classify
# resident waves per SIMD of the fp32 kernels (tools/kernel_resources.py: 256+80 / 246 / 256+256 registers per lane; the
# VGPR_Count column of the rocprofv3 CSV does not include the accumulator half reliably)
WAVES_PER_SIMD = {"oc_solve": 1, "oc_solve_wide": 1, "oc_solve_wide_w4": 1, "oc_solve_resume": 1, "oc_solve_seed_f32": 1, "aux_riccati": 2, "aux_forward": 1}


def code_object_registers(kernel_name):
    """"VGPRs+AGPRs" of a kernel as the COMPILER reports them (tools/kernel_resources.py <model> --json FILE, run at build time;
    LFSD_KERNEL_RESOURCES names the file).  The rocprofv3 CSV's VGPR_Count / Accum_VGPR_Count columns are allocation
    granules, not the kernel's footprint (round 3 reported 200 for a kernel the compiler builds with 256 + 137)."""
    import os
    import re
    path = os.environ.get("LFSD_KERNEL_RESOURCES")
    if not path or not os.path.exists(path):
        return None
    norm = lambda n: re.sub(r"^void", "", re.sub(r"\s+", "", re.sub(r"lfsd_gen_\w+::Model", "M", n)).split("(")[0])
    want = norm(kernel_name)
    for e in json.load(open(path)):
        if norm(e["name"]) == want:
            return {"vgprs": e.get("VGPRs"), "agprs": e.get("AGPRs"), "scratch_bytes_per_lane": e.get("ScratchSize"),
                    "lds_bytes_per_workgroup": e.get("LDS"), "occupancy_waves_per_simd": e.get("Occupancy")}
    return None


def main(argv):
    out = argv[1]
    res = {}
    for path in argv[2:]:                      # one pass per file: ratios only between counters of the SAME pass
        tot = defaultdict(lambda: defaultdict(float))
        regs = {}
        rows = []
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                key = classify(row["Kernel_Name"])
                if key in WAVES_PER_SIMD:
                    rows.append((key, row))
        gmax = defaultdict(int)
        for key, row in rows:
            gmax[key] = max(gmax[key], int(row["Grid_Size"]))
        nd = defaultdict(set)
        for key, row in rows:
            if int(row["Grid_Size"]) != gmax[key]:
                continue
            tot[key][row["Counter_Name"]] += float(row["Counter_Value"])
            nd[key].add(row["Dispatch_Id"])
            regs[key] = row["Kernel_Name"]
        for key, c in tot.items():
            wps = WAVES_PER_SIMD[key]
            r = res.setdefault(key, {"code_object": code_object_registers(regs.get(key, "")), "kernel_name": regs.get(key),
                                     "waves_per_simd": wps, "counters": {}})
            r["counters"].update({k: v / max(1, len(nd[key])) for k, v in c.items() if k not in r["counters"]})      # per launch
            r["full_batch_launches"] = len(nd[key])
            wc = c.get("SQ_WAVE_CYCLES")
            if wc and "SQ_ACTIVE_INST_VALU" in c:
                r["valu_issue_util"] = c["SQ_ACTIVE_INST_VALU"] / wc
            if wc and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                r["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc / wps)
            if wc and "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
                r["valu_mfma_coexec"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / (4.0 * wc / wps)
            if c.get("SQ_INSTS_VALU") and "SQ_INSTS_MFMA" in c:
                r["mfma_per_valu_inst"] = c["SQ_INSTS_MFMA"] / c["SQ_INSTS_VALU"]
            if c.get("SQ_INSTS_VALU") and "SQ_THREAD_CYCLES_VALU" in c:
                r["valu_lane_util"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"])
            if "SQ_INSTS_VALU_FMA_F64" in c:      # fp64 workloads (gpu_profile.sh F64_FLOPS=1): the same fold over the fp64 instruction classes
                n = max(1, len(nd[key]))
                r["valu_f64_insts_per_launch"] = {k2: c.get("SQ_INSTS_VALU_" + k2, 0.0) / n for k2 in ("FMA_F64", "ADD_F64", "MUL_F64", "TRANS_F64")}
                if "SQ_INSTS_VALU_FLOPS_FP64" in c and c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_INSTS_VALU"):
                    lu = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"])
                    r["SQ_INSTS_VALU_FLOPS_FP64_per_launch"] = c["SQ_INSTS_VALU_FLOPS_FP64"] / n
                    r["valu_flops64_executed_per_launch"] = c["SQ_INSTS_VALU_FLOPS_FP64"] / n * 64.0 * lu
            if "SQ_INSTS_VALU_FMA_F32" in c:
                n = max(1, len(nd[key]))
                r["valu_f32_insts_per_launch"] = {k2: c.get("SQ_INSTS_VALU_" + k2, 0.0) / n for k2 in ("FMA_F32", "ADD_F32", "MUL_F32", "TRANS_F32")}
                if "SQ_INSTS_VALU_FLOPS_FP32" in c:
                    # (= 2 FMA + ADD + MUL + TRANS per wave-instruction, packed ones at their double weight: checked on
                    #  the Riccati sweep, which has no packed instruction)
                    r["SQ_INSTS_VALU_FLOPS_FP32_per_launch"] = c["SQ_INSTS_VALU_FLOPS_FP32"] / n
                    if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_INSTS_VALU"):
                        lu = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"])
                        r["valu_flops_executed_per_launch"] = c["SQ_INSTS_VALU_FLOPS_FP32"] / n * 64.0 * lu
    import os
    res["collected_at_commit"] = os.environ.get("LFSD_COMMIT")
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    res.pop("collected_at_commit")
    print(json.dumps({k: {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "counters"}
                      for k, v in res.items()}))


if __name__ == "__main__":
    main(sys.argv)
