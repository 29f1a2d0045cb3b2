"""Post-build lint of the gfx950 code in libidocp_hip.so (advisor, round 5).

The in-register solves of dev_dense.hpp (choleskySolveRows, cholForwardFused ...) issue `v_fmac_f64_dpp` / `v_mov_b64_dpp ... row_newbcast:n`
from inline assembly.  Two hardware rules apply that the compiler's hazard recogniser does NOT enforce inside inline assembly:
  * a VGPR written by a VALU instruction may be read as the DPP operand (src0) only two wait states later;
  * the result of a transcendental-unit instruction (v_rcp / v_rsq / v_sqrt ...) may be read by a VALU instruction only one wait state later.
The spacing is kept by hand in the source (neighbouring statements, s_nop).  This test disassembles every code object of the built library and
checks both rules on the straight-line instruction stream, so a compiler or flag change that moves a producer next to its consumer fails here
instead of producing wrong numbers on the GPU."""
import glob
import os
import re
import shutil
import subprocess

import pytest

from idocp_amd import build

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_")
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)\b")


def regs(tok):
    """set of VGPR numbers named by one operand token (v7, v[4:5], -v[4:5], |v3| ...)"""
    m = REG.search(tok)
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def parse(line):
    """(mnemonic, [operand tokens]) of a disassembly line, or None"""
    code = line.split("//")[0].strip()
    if not code or code.endswith(":") or code.startswith("."):
        return None
    parts = code.split(None, 1)
    ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


def lint(path):
    return scan(subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", path], capture_output=True, text=True, check=True).stdout)


def scan(txt):
    bad, n_dpp, n_trans = [], 0, 0
    window = []          # the instructions that came before, newest last: (wait states it occupies, mnemonic, VGPRs it writes, is VALU)
    for line in txt.splitlines():
        if re.match(r"^[0-9a-f]+ <", line):      # a new function: no straight-line predecessor
            window = []
            continue
        p = parse(line)
        if p is None:
            continue
        mn, ops = p
        if mn == "s_nop":
            window.append((int(ops[0], 0) + 1, mn, set(), False))
            window = window[-4:]
            continue
        is_valu = mn.startswith("v_")
        dst = regs(ops[0]) if (is_valu and ops and not mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane"))) else set()
        srcs = set().union(*[regs(t) for t in ops[1:]]) if is_valu else set()
        if is_valu and mn.startswith(("v_cmp",)):
            srcs = set().union(*[regs(t) for t in ops])
        # rule 1: DPP operand (src0 = the operand that is permuted) at least two wait states after its VALU producer
        if "_dpp" in mn and "row_newbcast" in line:
            n_dpp += 1
            src0 = regs(ops[1]) if len(ops) > 1 else set()
            gap = 0
            for ws, pmn, pdst, pvalu in reversed(window):
                if gap >= 2:
                    break
                if pvalu and pdst & src0:
                    bad.append("DPP operand %s read %d wait state(s) after %s wrote it: %s" % (ops[1], gap, pmn, line.strip()[:110]))
                gap += ws
        # rule 2: the consumer of a transcendental result is not the very next VALU instruction
        if is_valu and window:
            ws, pmn, pdst, pvalu = window[-1]
            if pvalu and TRANS.match(pmn) and pdst & (srcs | (dst if "fmac" in mn else set())):
                bad.append("%s result read with no wait state: %s" % (pmn, line.strip()[:110]))
        if is_valu and TRANS.match(mn):
            n_trans += 1
        window.append((1, mn, dst, is_valu))
        window = window[-4:]
    return bad, n_dpp, n_trans


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump")
def test_dpp_and_trans_wait_states_in_the_built_library(tmp_path):
    lib = build.build_extension()
    work = tmp_path / "lib"
    work.mkdir()
    shutil.copy(lib, work / "libidocp_hip.so")      # (llvm-objdump --offloading extracts next to its input)
    subprocess.run([OBJDUMP, "--offloading", "libidocp_hip.so"], cwd=work, capture_output=True, check=True)
    objs = sorted(glob.glob(str(work / "*gfx950")))
    assert objs, "no gfx950 code object found in the library"
    bad, n_dpp, n_trans = [], 0, 0
    for o in objs:
        b, d, t = lint(o)
        bad += b; n_dpp += d; n_trans += t
    assert n_dpp > 1000, "the library should contain the DPP solves (found %d row_newbcast instructions)" % n_dpp
    assert not bad, "%d hazards:\n%s" % (len(bad), "\n".join(bad[:20]))


def test_the_lint_sees_a_hazard_when_there_is_one():
    ok = """
	v_mul_f64 v[28:29], v[2:3], v[30:31]
	s_nop 1
	v_fmac_f64_dpp v[26:27], -v[28:29], v[28:29] row_newbcast:1 row_mask:0xf bank_mask:0xf
	v_rsq_f64_e32 v[30:31], v[24:25]
	v_cmp_nlt_f64_e32 vcc, 0, v[24:25]
	v_mul_f64 v[32:33], -v[24:25], v[30:31]
"""
    assert scan(ok)[0] == []
    one_short = ok.replace("s_nop 1", "s_nop 0")
    assert len(scan(one_short)[0]) == 1 and "DPP operand" in scan(one_short)[0][0]
    adjacent = "\tv_mul_f64 v[28:29], v[2:3], v[30:31]\n\tv_mov_b64_dpp v[24:25], v[28:29] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
    assert len(scan(adjacent)[0]) == 1
    trans = "\tv_rcp_f64_e32 v[30:31], v[24:25]\n\tv_mul_f64 v[32:33], v[30:31], v[30:31]\n"
    assert len(scan(trans)[0]) == 1 and "no wait state" in scan(trans)[0][0]
    spaced = "\tv_rcp_f64_e32 v[30:31], v[24:25]\n\ts_nop 0\n\tv_mul_f64 v[32:33], v[30:31], v[30:31]\n"
    assert scan(spaced)[0] == []
