"""The build-time assembly check (lfsd_amd/isa_check.py): VGPR spills placed before the exec restore of a join block -- the pattern
behind two wrong builds of the wide OC kernel in round 6 (profiles/r06_v_spill_before_exec_restore.txt) -- on the assembly of the
wrong build itself (tests/golden/isa_spill_before_exec_restore.s), on constructed cases, and as part of runtime.build_library."""
import json
import os

import pytest

import lfsd_amd  # noqa: F401
from lfsd_amd import isa_check, models, runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fn(body):
    return "_Zkernel: ; @_Zkernel\n" + body + "\ts_endpgm\n.Lfunc_end0:\n"


REGION = """\
	v_cmp_gt_i32_e32 vcc, s16, v182
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_3
.LBB0_2:
	global_load_dword v5, v[0:1], off
	v_add_u32_e32 v4, 64, v4
	v_cmp_le_i32_e32 vcc, s16, v4
	s_or_b64 s[14:15], vcc, s[14:15]
	s_waitcnt vmcnt(0)
	global_store_dword v[2:3], v5, off
	s_andn2_b64 exec, exec, s[14:15]
	s_cbranch_execnz .LBB0_2
.LBB0_3:
"""


def test_the_wrong_build_of_round_6_is_flagged():
    text = open(os.path.join(ROOT, "tests", "golden", "isa_spill_before_exec_restore.s")).read()
    hz = isa_check.find_exec_hazards(text)
    assert len(hz) == 12 and {h["block"] for h in hz} == {".LBB4_487"} and {h["kind"] for h in hz} == {"spill"}
    assert any(h["instr"] == "v_accvgpr_write_b32 a12, v10" for h in hz)          # the cost of the accepted roll-out
    assert all("oc_solve_wide_kernel" in h["function"] for h in hz)
    # the same text with the exec restore where the compiler normally puts it -- at the top of the join block -- is clean
    lines = text.split("\n")
    k = next(i for i, ln in enumerate(lines) if ln.startswith(".LBB4_487:"))
    r = next(i for i in range(k, len(lines)) if lines[i].strip().startswith("s_or_b64 exec, exec"))
    fixed = lines[:k + 1] + [lines[r]] + lines[k + 1:r] + lines[r + 1:]
    assert isa_check.find_exec_hazards("\n".join(fixed)) == []


@pytest.mark.parametrize("before,after,kinds", [
    # SGPR spills do not depend on exec
    ("\tv_writelane_b32 v255, s44, 36\n\tv_readlane_b32 s2, v255, 3\n", "\tv_accvgpr_write_b32 a12, v10\n", []),
    # a VGPR spill to an AGPR / to scratch before the restore
    ("\tv_writelane_b32 v255, s44, 36\n\tv_accvgpr_write_b32 a12, v10\n\ts_waitcnt vmcnt(0)\n", "", ["spill"]),
    ("\tscratch_store_dwordx2 off, v[118:119], off offset:16 ; 8-byte Folded Spill\n", "", ["spill"]),
    # an ordinary store before the restore is the program's business
    ("\tscratch_store_dword off, v7, off offset:16\n\tglobal_store_dword v[2:3], v5, off\n", "", []),
    # a reload before the restore: a hazard when the register is read behind the restore ...
    ("\tv_accvgpr_read_b32 v7, a3\n", "\tv_add_f32_e32 v8, v7, v7\n", ["reload"]),
    ("\tscratch_load_dwordx2 v[6:7], off, off offset:208 ; 8-byte Folded Reload\n", "\tv_mul_f32_e32 v1, v7, v2\n", ["reload"]),
    ("\tv_accvgpr_read_b32 v7, a3\n", "\tv_fmac_f32_e32 v7, v1, v2\n", ["reload"]),            # (the destination of a mac is an operand)
    # ... and none when it is used inside the region only, or written first
    ("\tv_accvgpr_read_b32 v7, a3\n\tds_write_b32 v7, v9\n", "\tv_mov_b32_e32 v7, 0\n\tv_add_f32_e32 v8, v7, v7\n", []),
    ("\tv_accvgpr_read_b32 v7, a3\n", "\tv_accvgpr_read_b32 v7, a4\n\tv_add_f32_e32 v8, v7, v7\n", []),
])
def test_constructed_join_blocks(before, after, kinds):
    text = _fn(REGION + before + "\ts_or_b64 exec, exec, s[0:1]\n" + after)
    assert [h["kind"] for h in isa_check.find_exec_hazards(text)] == kinds


def test_only_join_blocks_are_examined():
    # the same spill inside the region (the block the lanes fall into, not the one the others jump to) is the allocator's right
    body = ("\ts_and_saveexec_b64 s[0:1], vcc\n\ts_cbranch_execz .LBB0_3\n.LBB0_2:\n\tv_accvgpr_write_b32 a12, v10\n"
            "\ts_or_b64 exec, exec, s[0:1]\n.LBB0_3:\n\ts_or_b64 exec, exec, s[0:1]\n\tv_accvgpr_write_b32 a13, v11\n")
    assert isa_check.find_exec_hazards(_fn(body)) == []
    # the entry of an else-region switches lanes on as well: a spill in front of it is flagged, one inside the else-region is not
    body = REGION + "\ts_andn2_saveexec_b64 s[0:1], s[0:1]\n\tv_accvgpr_write_b32 a12, v10\n\ts_or_b64 exec, exec, s[0:1]\n"
    assert isa_check.find_exec_hazards(_fn(body)) == []
    body = REGION + "\tv_accvgpr_write_b32 a12, v10\n\ts_or_saveexec_b64 s[0:1], s[0:1]\n\ts_xor_b64 exec, exec, s[0:1]\n"
    assert [h["kind"] for h in isa_check.find_exec_hazards(_fn(body))] == ["spill"]
    # any other write of exec ends the scan of a join block
    body = REGION + "\ts_mov_b64 exec, s[4:5]\n\tv_accvgpr_write_b32 a12, v10\n\ts_or_b64 exec, exec, s[0:1]\n"
    assert isa_check.find_exec_hazards(_fn(body)) == []
    assert isa_check.summary(_fn(REGION)) == {"functions": 1, "join_blocks": 1}
    # the exit of a divergent loop that no skip branch targets (a loop every lane group enters) is examined too
    loop = ("\ts_mov_b64 s[0:1], exec\n.LBB0_2:\n\tglobal_load_dword v5, v[0:1], off\n\ts_andn2_b64 exec, exec, s[14:15]\n\ts_cbranch_execnz .LBB0_2\n"
            "; %bb.3:\n\tv_accvgpr_write_b32 a12, v10\n\ts_or_b64 exec, exec, s[0:1]\n")
    assert [h["kind"] for h in isa_check.find_exec_hazards(_fn(loop))] == ["spill"]
    assert isa_check.find_exec_hazards(_fn(loop.replace("\tv_accvgpr_write_b32 a12, v10\n\ts_or_b64 exec, exec, s[0:1]\n",
                                                        "\ts_or_b64 exec, exec, s[0:1]\n\tv_accvgpr_write_b32 a12, v10\n"))) == []


def test_every_product_library_carries_a_clean_assembly_record():
    """__graft_entry__.build() goes through runtime.build_library: both translation units scanned, the record written next to the
    library and tied to its bytes (a library replaced by hand would not pass for checked)."""
    for kind in models.ZOO:
        spec = models.ZOO[kind]()[0].model_spec()
        lib = runtime.build_library(spec)
        assert runtime.isa_record_clean(lib), (kind, lib)
        rec = json.load(open(runtime.isa_record_path(lib)))
        for unit in ("capi", "riccati"):
            u = rec["units"][unit]
            assert u["hazards"] == 0 and u["functions"] >= 2 and u["join_blocks"] > 50, (kind, unit, u)
    # a record does not vouch for other bytes
    other = lib + ".copy"
    try:
        with open(lib, "rb") as f, open(other, "wb") as g:
            g.write(f.read() + b"\0")
        with open(runtime.isa_record_path(lib)) as f, open(runtime.isa_record_path(other), "w") as g:
            g.write(f.read())
        assert not runtime.isa_record_clean(other)
    finally:
        for p in (other, runtime.isa_record_path(other)):
            if os.path.exists(p):
                os.remove(p)


def test_a_flagged_unit_is_rebuilt_with_the_next_schedule_and_a_build_without_a_clean_one_fails(tmp_path, monkeypatch):
    spec = models.ZOO["pendulum"]()[0].model_spec()
    runtime.write_header(spec)
    real = isa_check.find_exec_hazards
    calls = []

    def first_capi_attempt_flagged(text):
        calls.append(len(text))
        if len(calls) == 1:
            return [{"function": "f", "block": ".LBB0_1", "line": 1, "kind": "spill", "instr": "v_accvgpr_write_b32 a0, v0"}]
        return real(text)

    monkeypatch.setattr(isa_check, "find_exec_hazards", first_capi_attempt_flagged)
    out = str(tmp_path / "lib.so")
    rec = runtime._checked_build(spec, out, what="test build of model")
    assert os.path.exists(out) and runtime.isa_record_clean(out)
    capi = rec["units"]["capi"]
    assert capi["flags"] == list(runtime.SCHEDULE_ALTERNATES[1]) and len(capi["rejected"]) == 1 and capi["rejected"][0]["hazards"] == 1
    assert rec["units"]["riccati"]["flags"] == [] and rec["units"]["riccati"]["rejected"] == []
    assert not [f for f in os.listdir(runtime.BUILD_DIR) if f.startswith("obj")]          # (the work directory is gone)
    # no clean schedule: no library
    monkeypatch.setattr(isa_check, "find_exec_hazards", lambda text: [{"function": "f", "block": ".LBB0_1", "line": 1, "kind": "spill", "instr": "x"}])
    monkeypatch.setattr(runtime, "SCHEDULE_ALTERNATES", ((),))
    out2 = str(tmp_path / "lib2.so")
    with pytest.raises(runtime.LfsdError, match="spills before an exec restore"):
        runtime._checked_build(spec, out2, what="test build of model")
    assert not os.path.exists(out2) and not os.path.exists(runtime.isa_record_path(out2))
