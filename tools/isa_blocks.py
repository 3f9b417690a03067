"""Per-basic-block instruction mix of one kernel in an isa_dump.py listing: python tools/isa_blocks.py <file.s> <regex on the demangled kernel name> <min instructions>"""
import re, sys, subprocess
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        d = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout
        if re.search(pat, d): start = i; print(d.strip()[:150]); break
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blk, blocks = "entry", {}
order = []
for l in lines[start+1:end]:
    m = re.match(r"^(\.LBB\w+):", l)
    if m: blk = m.group(1); continue
    if not l.startswith("\t") or l.startswith("\t.") or l.startswith("\t;"): continue
    if blk not in blocks: blocks[blk] = []; order.append(blk)
    blocks[blk].append(l.strip())
for b in order:
    ins = blocks[b]
    if len(ins) < int(sys.argv[3]) : continue
    c = lambda p: sum(1 for x in ins if re.match(p, x))
    br = [x for x in ins if x.startswith("s_cbranch") or x.startswith("s_branch")]
    print("%-12s n %5d valu %5d fma64 %4d ds_r %4d ds_w %4d scr_ld %4d scr_st %4d glb %4d wait %4d acc %4d  %s" % (b, len(ins), c(r"v_"), c(r"v_(fma|mul|add)_f64"), c(r"ds_read|ds_load"), c(r"ds_write|ds_store"), c(r"scratch_load"), c(r"scratch_store"), c(r"global_"), c(r"s_waitcnt"), c(r"v_accvgpr"), " ".join(x.split()[-1] for x in br)))
