"""Register / scratch / LDS usage of every kernel of a model library as the compiler reports it
(-Rpass-analysis=kernel-resource-usage).  usage: kernel_resources.py <model> [extra hipcc flags...]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfsd_amd
from lfsd_amd import models, runtime

kind = sys.argv[1]
extra = sys.argv[2:]
json_out = None
if "--json" in extra:            # kernel_resources.py <model> --json FILE: the same table as JSON (tools/issue_counters.py reads it)
    i = extra.index("--json")
    json_out = extra[i + 1]
    del extra[i:i + 2]
table = []
oc, _, _ = models.ZOO[kind]()
spec = oc.model_spec()
runtime.write_header(spec)
cmds, objs = runtime.hipcc_commands(spec, "/tmp/kres_%s.so" % spec.hash(), extra=["-Rpass-analysis=kernel-resource-usage"] + extra)
for cmd in cmds[:2]:
    r = subprocess.run(cmd, cwd=runtime.CSRC_DIR, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"lfsd_gen_\w+::Model", "M", name)[:110]
            vals = {}
        for key in ("VGPRs", "AGPRs", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]", "LDS Size \\[bytes/block\\]", "SGPRs"):
            m2 = re.search(r"remark: .*?%s: (\d+)" % key, line)
            if m2 and name:
                vals[key.split(" ")[0].replace("\\", "")] = int(m2.group(1))
                if key.startswith("LDS"):
                    print("%-112s %s" % (name, vals))
                    table.append(dict(name=name, **vals))
for o in objs:
    if os.path.exists(o):
        os.remove(o)
if json_out:
    import json
    with open(json_out, "w") as f:
        json.dump(table, f, indent=1)
