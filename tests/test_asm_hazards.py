"""CPU: the inline-asm hazard rules of csrc/conv_common.hpp (INLINE-ASM HAZARD CHECKLIST) are enforced on what was BUILT:
tools/check_asm_hazards.py disassembles every gfx950 code object of the product library and looks for a VALU-written SGPR
consumed by a vector-memory instruction within five wait states (H1), an LDS-DMA right behind a write of M0 (H2) and a wide
store whose data registers are overwritten within two wait states (H3).  The checker itself is pinned on synthetic streams
(each rule must fire and must not fire), and the sources are grepped: raw vector-memory asm only inside the wrapper block."""
import os
import re
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_asm_hazards as chk  # noqa: E402

CSRC = os.path.join(ROOT, "multigriddet_amd", "csrc")


def test_checker_fires_on_each_rule_and_only_then():
    bad1 = ["0000000000001000 <k>:", "\tv_readfirstlane_b32 s4, v1   // 0: X", "\ts_nop 2   // 4: X",
            "\tbuffer_load_dwordx4 v[0:3], v5, s[4:7], 0 offen   // 8: X"]
    assert any(" H1 " in f for f in chk.check_lines("t", bad1))
    ok1 = [bad1[0], bad1[1], "\ts_nop 4   // 4: X", bad1[3]]
    assert not chk.check_lines("t", ok1)
    bad1b = [bad1[0], "\tv_readfirstlane_b32 s9, v1   // 0: X", "\tglobal_load_dwordx4 v[0:3], v5, s[8:9] offset:16   // 8: X"]
    assert any(" H1 " in f for f in chk.check_lines("t", bad1b))
    bad2 = [bad1[0], "\ts_mov_b32 m0, s3   // 0: X", "\tbuffer_load_dwordx4 v5, s[40:43], 0 offen lds   // 4: X"]
    assert any(" H2 " in f for f in chk.check_lines("t", bad2))
    ok2 = [bad1[0], bad2[1], "\ts_nop 0   // 4: X", bad2[2]]
    assert not chk.check_lines("t", ok2)
    bad3 = [bad1[0], "\tglobal_store_dwordx4 v[10:11], v[4:7], off sc0 sc1   // 0: X", "\tv_mov_b32_e32 v5, v20   // 4: X"]
    assert any(" H3 " in f for f in chk.check_lines("t", bad3))
    ok3 = [bad1[0], bad3[1], "\ts_nop 1   // 4: X", bad3[2]]
    assert not chk.check_lines("t", ok3)
    ok3b = [bad1[0], bad3[1], "\tv_mov_b32_e32 v9, v20   // 4: X"]          # another register: fine
    assert not chk.check_lines("t", ok3b)


def test_product_library_disassembly_is_clean():
    from multigriddet_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    seen = {"vmem": 0, "dma": 0, "objects": 0}
    findings = []
    for name, lines in chk.disassemble(_lib.LIB_PATH):
        seen["objects"] += 1
        seen["vmem"] += sum(1 for l in lines if re.match(r"\s+(buffer_|global_)", l))
        seen["dma"] += sum(1 for l in lines if re.search(r"buffer_load_dwordx4 .* lds", l))
        findings += chk.check_lines(name, lines)
    assert seen["objects"] >= 8 and seen["vmem"] > 2000 and seen["dma"] > 200, seen     # the scan really saw the kernels
    assert not findings, "\n".join(findings[:20])


def test_raw_vector_memory_asm_lives_in_the_wrapper_block_only():
    """No `asm volatile("global_... / buffer_...")` and no use of the DMA string macros outside conv_common.hpp's wrapper block
    in the product sources (csrc/diag/ is the diagnostic library)."""
    pat = re.compile(r'asm\s+volatile\s*\(\s*(MGD_DMA_|MGD_Q_|"[^"]*(global_|buffer_|flat_))')
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hip", ".cpp")):
            continue
        src = open(os.path.join(CSRC, f)).read()
        assert not pat.search(src), f
    hdr = open(os.path.join(CSRC, "conv_common.hpp")).read()
    i0, i1 = hdr.index("// Inline-asm wrappers."), hdr.index("// ====", hdr.index("// Inline-asm wrappers."))
    outside = hdr[:i0] + hdr[i1:]
    assert not pat.search(outside)
    assert pat.search(hdr[i0:i1]) and "INLINE-ASM HAZARD CHECKLIST" in hdr[i0:i1]
