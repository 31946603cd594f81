"""The MFMA operand-overwrite rule as a build invariant: tools/isa_lint.py over the gfx950 code objects of
the shipped library.  CPU suite: hipcc cross-compiles here and llvm-objdump reads the result; nothing runs on a GPU."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
isa_lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_lint)

CSRC = os.path.join(ROOT, "se3_equi_graph_registration_amd", "csrc")
needs_llvm = pytest.mark.skipif(not os.path.exists(os.path.join(isa_lint.LLVM, "llvm-objdump")), reason="ROCm LLVM tools not installed")


def test_parser_and_rule_on_a_synthetic_listing():
    """The rule itself, on hand-written listings: a load into the A operand 3 cycles behind a 128-bit-operand MFMA is a violation, the same
    load behind `s_nop 7` is not; the path through a backward branch is followed; single-register fp32 operands are only counted."""
    listing = """
0000000000001000 <k_bad>:
\tv_mfma_f32_16x16x32_bf16 v[20:23], v[86:89], v[60:63], v[20:23]// 000000001000: D3B50014 04527956
\ts_nop 1                                                    // 000000001008: BF800001
\tds_read_b128 v[86:89], v31 offset:19072                    // 00000000100C: D9FE4A80 5600001F
\ts_endpgm                                                   // 000000001014: BF810000
0000000000002000 <k_good>:
\tv_mfma_f32_16x16x32_bf16 v[20:23], v[86:89], v[60:63], v[20:23]// 000000002000: D3B50014 04527956
\ts_nop 7                                                    // 000000002008: BF800007
\tds_read_b128 v[86:89], v31 offset:19072                    // 00000000200C: D9FE4A80 5600001F
\ts_endpgm                                                   // 000000002014: BF810000
0000000000003000 <k_loop>:
\tglobal_load_dwordx4 v[60:63], v[2:3], off                  // 000000003000: DC5C8000 3C7F0002
\ts_waitcnt vmcnt(0)                                         // 000000003008: BF8C0F70
\tv_mfma_f32_16x16x32_bf16 v[20:23], v[86:89], v[60:63], v[20:23]// 00000000300C: D3B50014 04527956
\ts_cbranch_scc1 65531                                       // 000000003014: BF85FFFB <k_loop+0x0>
\ts_endpgm                                                   // 000000003018: BF810000
0000000000004000 <k_fp32>:
\tv_mfma_f32_16x16x4_f32 v[14:17], v114, v117, v[14:17]      // 000000004000: D3C5000E 043AEB72
\tds_read_b32 v114, v223 offset:8624                         // 000000004008: D86C21B0 720000DF
\ts_endpgm                                                   // 000000004010: BF810000
"""
    ks = isa_lint.parse(listing)
    assert set(ks) == {"k_bad", "k_good", "k_loop", "k_fp32"}
    v, n_multi, n_single, _, _ = isa_lint.lint_kernel("k_bad", ks["k_bad"], 8)
    assert n_multi == 1 and len(v) == 1 and v[0]["issue_cycles_between"] == 2 and v[0]["overlap"] == ["v86", "v87", "v88", "v89"]
    v, n_multi, _, _, _ = isa_lint.lint_kernel("k_good", ks["k_good"], 8)
    assert n_multi == 1 and v == []
    v, _, _, _, _ = isa_lint.lint_kernel("k_loop", ks["k_loop"], 8)
    assert len(v) == 1 and v[0]["load"].startswith("global_load_dwordx4 v[60:63]")           # found through the backward branch
    v, n_multi, n_single, n_d1, _ = isa_lint.lint_kernel("k_fp32", ks["k_fp32"], 8)
    assert v == [] and n_multi == 0 and n_single == 1 and n_d1 == 1
    assert isa_lint.regs("v[86:89]") == {("v", 86), ("v", 87), ("v", 88), ("v", 89)} and isa_lint.regs("a3") == {("a", 3)} and not isa_lint.regs("s[4:5]")


@needs_llvm
def test_shipped_code_objects_keep_mfma_operands_until_the_group_has_issued():
    """Every gfx950 code object of the library: no VMEM / LDS load writes an A, B or C source register of a multi-register-operand MFMA
    within 8 issue cycles of it (the closed groups of csrc/egnn_mfma16.h: sched_barrier, operands kept alive over `s_nop 7`, fragment
    reads issued before the first matrix instruction).  The split products must actually be there (>= 18 kernels), or the lint checks nothing."""
    objs = isa_lint.default_objects()
    assert objs, "build the library first (__graft_entry__.build())"
    reports = isa_lint.run(objs, 8)
    assert reports, "no gfx950 code objects found"
    violations = [v for r in reports for v in r["violations"]]
    assert not violations, violations[:5]
    multi = {(r["file"], k): v for r in reports for k, v in r["kernels"].items() if v["mfma_multi_operand"]}
    assert len(multi) >= 18 and sum(v["mfma_multi_operand"] for v in multi.values()) >= 1000, len(multi)
    margins = [v["min_cycles_to_overwriting_load"] for v in multi.values() if v["min_cycles_to_overwriting_load"] is not None]
    assert margins and min(margins) >= 8, min(margins)
