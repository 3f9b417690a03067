"""Device assembly of a model library's two translation units (no GPU needed).
usage: isa_dump.py <model> <outdir> [extra hipcc flags...]   ->  <outdir>/<model>_capi.s, <model>_riccati.s
Prints, per kernel, the count of a few instruction classes (s_barrier, v_mfma, scratch, ds_*, v_pk_*)."""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfsd_amd  # noqa: F401
from lfsd_amd import models, runtime

kind, outdir, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
os.makedirs(outdir, exist_ok=True)
oc, _, _ = models.ZOO[kind]()
spec = oc.model_spec()
runtime.write_header(spec)
cmds, objs = runtime.hipcc_commands(spec, "/tmp/isa_%s.so" % spec.hash(), extra=extra)
for cmd, unit in zip(cmds[:2], ("capi", "riccati")):
    out = os.path.join(outdir, "%s_%s.s" % (kind, unit))
    c = [a for a in cmd if a != "-c"]
    c[c.index("-o") + 1] = out
    c += ["-S", "--cuda-device-only"]
    r = subprocess.run(c, cwd=runtime.CSRC_DIR, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    name, counts = None, {}
    pats = {"s_barrier": r"\ts_barrier", "mfma": r"\tv_mfma", "scratch": r"\tscratch_|\tbuffer_(load|store).*offen", "ds": r"\tds_",
            "v_pk": r"\tv_pk_", "valu": r"\tv_", "waitcnt": r"\ts_waitcnt"}
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"lfsd_gen_\w+::Model", "M", name)[:100]
            counts[name] = {k: 0 for k in pats}
            continue
        if name:
            for k, p in pats.items():
                if re.match(p, line):
                    counts[name][k] += 1
    for k, v in counts.items():
        if "kernel" in k:
            print("%-6s %-100s %s" % (unit, k, v))
