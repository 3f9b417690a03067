#!/usr/bin/env python3
"""Fold rocprofv3 `--pmc` SQ passes into profiles/<round>_issue_counters.json (bench.py reads `valu_issue_util` and
`mfma_busy` of the dominant kernel for its `roofline` object).

usage: tools/issue_counters.py OUT.json PASS1_counter_collection.csv [PASS2_counter_collection.csv ...]

Units (MI355X_MICROARCH.md, per-instruction cycle constants): SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles summed
over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs.
  valu_issue_util = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES      share of a resident wave's lifetime it spends issuing VALU work
  mfma_busy       = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES / waves_per_simd)   share of the SIMD-time covered by
                    waves during which the matrix pipe is busy (waves_per_simd from the kernel's register footprint)
"""
import csv
import json
import sys
from collections import defaultdict

KERNELS = {"oc_solve": "oc_solve_kernel", "aux_riccati": "aux_riccati_kernel", "aux_forward": "aux_forward_kernel"}
# resident waves per SIMD of the fp32 kernels (tools/kernel_resources.py: 256+80 / 246 / 256+256 registers per lane; the
# VGPR_Count column of the rocprofv3 CSV does not include the accumulator half reliably)
WAVES_PER_SIMD = {"oc_solve": 1, "aux_riccati": 2, "aux_forward": 1}


def main(argv):
    out = argv[1]
    res = {}
    for path in argv[2:]:                      # one pass per file: ratios only between counters of the SAME pass
        tot = defaultdict(lambda: defaultdict(float))
        regs = {}
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                for key, pat in KERNELS.items():
                    if pat in row["Kernel_Name"]:
                        tot[key][row["Counter_Name"]] += float(row["Counter_Value"])
                        v = int(row["VGPR_Count"]) + int(row["Accum_VGPR_Count"])
                        regs[key] = max(regs.get(key, 0), v)
        for key, c in tot.items():
            wps = WAVES_PER_SIMD[key]
            r = res.setdefault(key, {"registers_per_lane": regs.get(key), "waves_per_simd": wps, "counters": {}})
            r["counters"].update({k: v for k, v in c.items() if k not in r["counters"]})
            wc = c.get("SQ_WAVE_CYCLES")
            if wc and "SQ_ACTIVE_INST_VALU" in c:
                r["valu_issue_util"] = c["SQ_ACTIVE_INST_VALU"] / wc
            if wc and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                r["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc / wps)
            if wc and "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
                r["valu_mfma_coexec"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / (4.0 * wc / wps)
            if c.get("SQ_INSTS_VALU") and "SQ_INSTS_MFMA" in c:
                r["mfma_per_valu_inst"] = c["SQ_INSTS_MFMA"] / c["SQ_INSTS_VALU"]
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "counters"}
                      for k, v in res.items()}))


if __name__ == "__main__":
    main(sys.argv)
