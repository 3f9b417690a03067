"""Build-time check of the gfx950 assembly hipcc produced for a model library (lfsd_amd.runtime.build_library).

What it looks for.  Round 6 found two builds of the wide OC kernel that were WRONG with sources the emulator, the sanitizers and
every other build of the same text agreed on (profiles/r06_f_wide_stale_cost.txt, r06_v_spill_before_exec_restore.txt):

    s_and_saveexec_b64 s[0:1], vcc        ; lanes b < n of a copy loop
    s_cbranch_execz .LBB4_487
    ...                                   ; the loop; exec shrinks as lanes finish
  .LBB4_487:                              ; JOIN block
    v_accvgpr_write_b32 a12, v10          ; <- the register allocator's spill of a value every lane needs later ...
    v_accvgpr_write_b32 a1, v178
    ...
    s_or_b64 exec, exec, s[0:1]           ; <- ... placed BEFORE the lanes are switched back on

The spill stores only the lanes that were active in the region; the others keep what the spill slot held before (here: the cost
of the previous iterate in every lane >= n -- a uniform value that later `v_cmp` + `s_cbranch_vccnz` pairs read from all lanes).
The compiler keeps such spills behind the exec restore unless the join block begins with SGPR-spill lane writes (`v_writelane`),
which both wrong builds had.  Nothing in the source is wrong and nothing in the source controls it; whether it happens changes with
any perturbation of the kernel (a printf, a clock read, another instruction scheduler).

So every build is scanned: in a block that is the target of an `s_cbranch_execz` (a join block) or that follows the back edge of a
divergent loop (`s_cbranch_execnz`: the loop's exit), between the label and the
`s_or_b64 exec, exec, ...` (or `s_or_saveexec` / `s_andn2_saveexec` / `s_xor_b64 exec`: the entry of an else-region) that re-enables lanes,
  * a VGPR spill store (`v_accvgpr_write_b32 aN, vM`, `scratch_store_* ; ... Folded Spill`) is a HAZARD;
  * a spill reload (`v_accvgpr_read_b32`, `scratch_load_* ; ... Folded Reload`) is a hazard when the reloaded register is read
    after the restore before it is written again (inactive lanes would read what the register held before).
`v_writelane_b32` / `v_readlane_b32` (SGPR spills) do not depend on exec and are fine there.

Pure text processing: no GPU, no toolchain.
"""
import re

_FUNC = re.compile(r"^([A-Za-z_][\w$.]*):")
_BLOCK = re.compile(r"^(\.LBB\d+_\d+):")
_EXECZ = re.compile(r"^\s+s_cbranch_execz\s+(\.LBB\d+_\d+)")
_EXECNZ = re.compile(r"^\s+s_cbranch_execnz\s+(\.LBB\d+_\d+)")
# what switches lanes back on at the top of a join block: the plain restore, and the entry of an else-region / of the next round of a
# waterfall (s_or_saveexec, s_andn2_saveexec, s_xor exec): a spill in front of any of them stores the lanes of the region just left
_RESTORE = re.compile(r"^(s_or_b64\s+exec,\s*exec,|s_or_saveexec_b64|s_andn2_saveexec_b64|s_xor_b64\s+exec,\s*exec,)")
_SPILL_STORE = re.compile(r"^(v_accvgpr_write_b32\s+a\d+,\s*v\d+|scratch_store_\w+\s.*Folded Spill)")
_SPILL_LOAD = re.compile(r"^(v_accvgpr_read_b32\s+(v\d+),\s*a\d+|scratch_load_(\w+)\s+(v\d+|v\[\d+:\d+\]),.*Folded Reload)")
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_EXEC_WRITE = re.compile(r"^s_\w*saveexec|^s_\w+\s+exec\b|^s_mov_b64\s+exec|^s_cmov\w*\s+exec")


def _regs(text):
    out = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _operands(instr):
    """(registers written, registers read) of a vector instruction line, by position: the first operand is the destination of
    every instruction this scan meets behind an exec restore (stores have none).  Conservative where it cannot tell."""
    body = instr.split(";")[0]
    parts = body.split(None, 1)
    if len(parts) < 2:
        return set(), set()
    op, args = parts
    ops = [a.strip() for a in args.split(",")]
    if op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "flat_store", "s_")) or op.startswith("v_cmp") and not op.endswith("_e64"):
        return set(), _regs(args)
    # v_cmp*_e64 writes an SGPR pair; v_readlane writes an SGPR: the first operand holds no VGPR then
    if "mac" in op or op.startswith(("v_writelane", "v_dot", "v_mfma", "v_smfmac")):      # the destination is an operand too
        return _regs(ops[0]), _regs(args)
    return _regs(ops[0]), _regs(",".join(ops[1:]))


def find_exec_hazards(asm_text):
    """-> list of {"function", "block", "line", "kind": "spill"|"reload", "instr"} for the pattern in the module docstring."""
    lines = asm_text.split("\n")
    hazards = []
    func, targets = None, set()
    # execz targets per function (labels are unique per function: .LBB<function index>_<n>)
    for ln in lines:
        m = _EXECZ.match(ln)
        if m:
            targets.add(m.group(1))
    n = len(lines)
    for i, ln in enumerate(lines):
        m = _FUNC.match(ln)
        if m and not ln.startswith(".L"):
            func = m.group(1)
            continue
        m = _BLOCK.match(ln)
        if m and m.group(1) in targets:
            label = m.group(1)
        elif _EXECNZ.match(ln):      # the fall-through behind the back edge of a divergent loop is its exit whether or not a skip branch targets it
            label = "behind line %d" % (i + 1)
        else:
            continue
        j, stores, loads, restore_at = i + 1, [], [], -1
        while j < n and (not lines[j].strip() or lines[j].strip().startswith(";") or (label.startswith("behind") and _BLOCK.match(lines[j]))):
            if _BLOCK.match(lines[j]) and _BLOCK.match(lines[j]).group(1) in targets:
                break                # (that block is examined under its own label)
            j += 1
        if j < n and _BLOCK.match(lines[j]) and label.startswith("behind"):
            continue
        while j < n:
            t = lines[j].strip()
            if not t or t.startswith(";"):
                j += 1
                continue
            if t.startswith(".L") or t.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                break
            if _RESTORE.match(t):
                restore_at = j
                break
            if _EXEC_WRITE.match(t):
                break
            if _SPILL_STORE.match(t):
                stores.append((j + 1, t))
            else:
                ml = _SPILL_LOAD.match(t)
                if ml:
                    loads.append((j + 1, t, _regs(ml.group(2) or ml.group(4))))
            j += 1
        if restore_at < 0:
            continue
        for at, t in stores:
            hazards.append({"function": func, "block": label, "line": at, "kind": "spill", "instr": t})
        for at, t, regs in loads:
            # the reloaded registers: read behind the restore before they are written again?
            pending, k = set(regs), restore_at + 1
            bad = False
            while k < n and pending:
                u = lines[k].strip()
                if not u or u.startswith(";"):
                    k += 1
                    continue
                if u.startswith(".L") or u.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                    break
                w, r = _operands(u)
                if r & pending:
                    bad = True
                    break
                pending -= w
                k += 1
            if bad:
                hazards.append({"function": func, "block": label, "line": at, "kind": "reload", "instr": t})
    return hazards


def summary(asm_text):
    """Counts for the build record: functions, join blocks examined."""
    targets = set(m.group(1) for m in (_EXECZ.match(ln) for ln in asm_text.split("\n")) if m)
    nfunc = sum(1 for ln in asm_text.split("\n") if ln.startswith(".Lfunc_end"))
    return {"functions": nfunc, "join_blocks": len(targets)}
