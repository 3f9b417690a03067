"""Scan the gfx950 assembly of a model library for VGPR spills placed before the exec restore of a join block
(lfsd_amd/isa_check.py; profiles/r06_v_spill_before_exec_restore.txt).  No GPU needed.

    python tools/isa_hazards.py <model> [extra hipcc flags ...]      compiles both translation units to assembly (tools/isa_dump.py) and scans
    python tools/isa_hazards.py --file <x.s> [...]                   scans assembly that is already there
    python tools/isa_hazards.py --records                            what runtime.build_library recorded for the libraries in csrc/build/
"""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfsd_amd  # noqa: F401
from lfsd_amd import isa_check, runtime


def report(path):
    text = open(path).read()
    hz = isa_check.find_exec_hazards(text)
    print("%s: %s, %d hazard(s)" % (path, isa_check.summary(text), len(hz)))
    for h in hz:
        name = subprocess.run(["c++filt", h["function"] or ""], capture_output=True, text=True).stdout.strip()[:110]
        print("    %-6s %s line %d  %s   | %s" % (h["kind"], h["block"], h["line"], h["instr"][:60], name))
    return len(hz)


if __name__ == "__main__":
    if sys.argv[1] == "--records":
        for f in sorted(os.listdir(runtime.BUILD_DIR)):
            if f.endswith(".isa.json"):
                rec = json.load(open(os.path.join(runtime.BUILD_DIR, f)))
                lib = os.path.join(runtime.BUILD_DIR, f[:-len(".isa.json")])
                print(f[:-len(".isa.json")], "clean" if runtime.isa_record_clean(lib) else "NOT CLEAN / not this file's record",
                      {u: (v["flags"], "rejected %d" % len(v["rejected"])) for u, v in rec["units"].items()})
        sys.exit(0)
    if sys.argv[1] == "--file":
        sys.exit(1 if sum(report(p) for p in sys.argv[2:]) else 0)
    out = tempfile.mkdtemp(prefix="isa_")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "isa_dump.py"), sys.argv[1], out] + sys.argv[2:],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    n = sum(report(os.path.join(out, f)) for f in sorted(os.listdir(out)) if f.endswith(".s"))
    sys.exit(1 if n else 0)
