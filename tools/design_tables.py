"""Markdown rows of DESIGN.md section 4 from a committed evidence set:  python tools/design_tables.py [prefix=profiles/r05]
(bench line, rocprofv3 kernel stats, the two folds of the PMC passes; nothing is measured here)."""
import csv, json, sys

pre = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05"
b = json.load(open(pre + "_bench.json"))
hbm = json.load(open(pre + "_hbm_traffic.json"))
iss = json.load(open(pre + "_issue_counters.json"))
stats = {}
for r in csv.DictReader(open(pre + "_kernel_stats.csv")):
    stats[r["Name"]] = r
def prof_ms(sub, f64=False):
    best = None
    for name, r in stats.items():
        if sub in name and (("double" in name) == f64):
            ms = float(r["AverageNs"]) / 1e6
            if best is None or float(r["TotalDurationNs"]) > best[1]: best = (ms, float(r["TotalDurationNs"]), int(r["Calls"]))
    return best
km = b["config"]["kernel_ms"]
B = b["config"].get("batch_per_gpu") or b["config"].get("batch") or 4096
alg = {"oc_solve": b["roofline"]["algorithmic_bytes_per_launch"]}
print("bench line: %.0f it/s, %.3f ms/step, kernels %s" % (b["value"], b["ms_per_step"], km))
print("| kernel | ms / launch (HIP events; rocprofv3 avg, calls) | memory-side traffic / launch (PMC) | executed vector TFLOP/s | VALU issue | EXEC lanes / instruction | SQ_WAIT_ANY / wave cycles | SQ_INSTS_VALU / launch | registers, scratch, LDS |")
print("|---|---|---|---|---|---|---|---|---|")
for k, sub in (("oc_solve", "oc_solve_kernel"), ("aux_riccati", "aux_riccati_kernel"), ("aux_forward", "aux_forward_kernel")):
    i = iss[k]; h = hbm[k]; c = i["counters"]; co = i.get("code_object", {})
    p = prof_ms(sub, f64=("double" in i["kernel_name"]))
    fl = i.get("valu_flops_executed_per_launch") or i.get("valu_flops64_executed_per_launch") or 0.0
    ms = km[k]
    print("| `%s` | %.3f (%s) | %.3f GB (%.2f fetched + %.2f written) | %.1f | %.0f %% x %d wave(s) | %.2f | %.0f %% | %.0f M | %s + %s, %s B, %.1f KB |" % (
        sub, ms, "%.3f, %d" % (p[0], p[2]) if p else "-", h["hbm_bytes_per_launch"] / 1e9, h["FETCH_SIZE_KB_per_launch"] * 1024 / 1e9, h["WRITE_SIZE_KB_per_launch"] * 1024 / 1e9,
        fl / (ms * 1e-3) / 1e12, 100 * i["valu_issue_util"], i["waves_per_simd"], i["valu_lane_util"], 100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_INSTS_VALU"] / 1e6,
        co.get("vgprs"), co.get("agprs"), co.get("scratch_bytes_per_lane"), (co.get("lds_bytes_per_workgroup") or 0) / 1024.0))
r = b["roofline"]
print("roofline object: achieved %.1f GB/s of %.0f (frac %.2e), traffic %s, valu_useful %.1f TFLOP/s (%.1f %%), source %s" % (
    r["achieved"], r["peak"], r["frac"], r.get("traffic"), r["valu_useful_tflops"], 100 * r["valu_frac"], (r.get("source") or "")[:200] if isinstance(r.get("source"), str) else r.get("source")))
