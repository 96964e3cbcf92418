"""Build-time check of the inline-asm hazard rules (conv_common.hpp, INLINE-ASM HAZARD CHECKLIST) on the DISASSEMBLY of the
product library: llvm-objdump of every gfx950 code object of libmgd_hip.so, scanned per straight-line run for
  H1  a VALU write of an SGPR (v_readfirstlane / v_readlane / v_cmp ... with an SGPR destination) read by a vector-memory
      instruction (descriptor, scalar offset or base) fewer than five wait states later;
  H2  a write of M0 (s_mov / s_add) with no wait state before the LDS-DMA that uses it;
  H3  a store of more than 64 bits whose data registers are written by a VALU instruction fewer than two wait states later.
One instruction = one wait state, `s_nop N` = N + 1; a branch target or a branch ends the run (conservative: states are not
carried across labels).  Exit code 1 and one line per finding; importable: check_library(path) -> list of findings.
usage: python3 tools/check_asm_hazards.py [libmgd_hip.so]"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_)")
SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _regs(rx, text):
    out = set()
    for m in rx.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def disassemble(lib):
    """Yield (code object name, [instruction lines]) for every gfx950 bundle of `lib`."""
    tmp = tempfile.mkdtemp(prefix="mgd_asm_")
    try:
        dst = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, dst)
        subprocess.run([OBJDUMP, "--offloading", dst], cwd=tmp, check=True, capture_output=True)
        for co in sorted(glob.glob(dst + ".*gfx950*")):
            txt = subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
            yield os.path.basename(co), txt.splitlines()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check_lines(name, lines):
    findings = []
    func = "?"
    sgpr_age = {}      # SGPR index -> wait states since a VALU wrote it
    m0_age = None      # wait states since M0 was written (None: long ago)
    stores = []        # [data VGPR set, wait states since the store, text]
    for raw in lines:
        line = raw.split("//")[0].strip()
        if not line:
            continue
        if line.endswith(":"):                         # label / function start: a new straight-line run
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m and not m.group(1).startswith("L"):
                func = m.group(1)
            sgpr_age, m0_age, stores = {}, None, []
            continue
        parts = line.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        states = 1
        if op == "s_nop":
            states = int(args, 0) + 1
        # ---- consumers first (the instruction sees the state BEFORE it)
        if VMEM.match(op):
            ops_ = [a.strip() for a in args.split(",")]
            # scalar operands of a vector-memory instruction: everything that is an SGPR (descriptor, soffset, saddr)
            used = _regs(SREG, args)
            for r in used:
                if r in sgpr_age and sgpr_age[r] < 5:
                    findings.append(f"{name}:{func}: H1 {op} reads s{r} {sgpr_age[r]} wait state(s) after a VALU wrote it: {line}")
            if " lds" in (" " + args) and m0_age is not None and m0_age < 1:
                findings.append(f"{name}:{func}: H2 LDS-DMA right behind a write of M0: {line}")
            if re.match(r"^(global|buffer|flat|scratch)_store_dwordx[34]", op):
                data = _regs(VREG, ops_[1] if op.startswith(("global", "flat", "scratch")) else ops_[0])
                stores.append([data, -1, line])
        if op.startswith("v_") and not op.startswith("v_cmp") and stores:
            dst = _regs(VREG, args.split(",")[0])
            for data, age, text in stores:
                if 0 <= age < 2 and dst & data:
                    findings.append(f"{name}:{func}: H3 {op} writes data registers of `{text}` {age} wait state(s) after it")
        # ---- producers
        if op.startswith(("v_readfirstlane", "v_readlane")) or (op.startswith("v_cmp") and args.strip().startswith("s")):
            for r in _regs(SREG, args.split(",")[0]):
                sgpr_age[r] = -1                       # becomes 0 below: the next instruction sees 0 wait states
        if op.startswith("s_") and re.match(r"^m0\b", args.strip()) and op not in ("s_nop",):
            m0_age = -1
        # ---- age everything by this instruction's wait states
        for r in list(sgpr_age):
            sgpr_age[r] += states
            if sgpr_age[r] > 8:
                del sgpr_age[r]
        if m0_age is not None:
            m0_age += states
            if m0_age > 4:
                m0_age = None
        for s in stores:
            s[1] += states
        stores = [s for s in stores if s[1] < 4]
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
            sgpr_age, m0_age, stores = {}, None, []
    return findings


def check_library(lib):
    out, n = [], 0
    for name, lines in disassemble(lib):
        n += 1
        out += check_lines(name, lines)
    if n == 0:
        out.append(f"{lib}: no gfx950 code object found")
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "multigriddet_amd", "csrc", "libmgd_hip.so")
    f = check_library(lib)
    for x in f:
        print(x)
    print(f"{lib}: {len(f)} finding(s)")
    sys.exit(1 if f else 0)
