"""Scans the gfx950 ISA the build generated (csrc/build/*-gfx950.s, kept by -save-temps) for hazards the compiler
cannot see because they sit inside inline-asm strings.

gfx940 family (gfx950 included): a VMEM store wider than 64 bits needs 2 wait states before a VALU instruction
rewrites its data registers.  Every 16-byte streaming store of the step path goes through `st_through(u32x4*)`
(gvec_device.hpp), an asm string the hazard recognizer does not parse; its trailing `s_nop 1` is the protection
and this test is the proof that no store site lost it."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "generalsreinforcementlearning_amd", "csrc", "build")


def _instructions(path):
    out = []
    for line in open(path):
        s = line.split(";")[0].strip()
        if not s or s.startswith((".", "//")) or s.endswith(":"):
            continue
        out.append(s)
    return out


@pytest.mark.parametrize("unit", ["gvec_kernels", "gvec_api"])
def test_wide_asm_stores_keep_their_wait_states(unit):
    path = os.path.join(BUILD, f"{unit}-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(path):
        import sys
        sys.path.insert(0, ROOT)
        from generalsreinforcementlearning_amd.csrc import build as B
        B.build(force=True, verbose=False)
    ins = _instructions(path)
    sites = [i for i, s in enumerate(ins) if re.match(r"global_store_dwordx[34]\b.*\bsc1\b", s)]
    if unit == "gvec_kernels":
        assert len(sites) > 100, "the step path's staged 16-byte stores were not found: has st_through changed its spelling?"
    for i in sites:
        nxt = ins[i + 1]
        m = re.match(r"s_nop\s+(\d+)", nxt)
        if m and int(m.group(1)) >= 1:
            continue
        # without the nop: the next two instructions must not be VALU writes of the store's data registers
        regs = re.search(r"v\[(\d+):(\d+)\]", ins[i].split(",")[1])
        lo, hi = int(regs.group(1)), int(regs.group(2))
        for k in (1, 2):
            dst = re.match(r"v_\w+\s+v(?:\[(\d+):(\d+)\]|(\d+))", ins[i + k])
            if dst:
                a = int(dst.group(1) or dst.group(3))
                b = int(dst.group(2) or dst.group(3))
                assert b < lo or a > hi, f"{unit}: `{ins[i]}` is followed within 2 wait states by `{ins[i + k]}`"


def test_no_cross_lane_read_behind_a_short_circuit():
    """ds_bpermute / DPP read zeros from lanes that are masked off in EXEC.  A `cond && gather(...)` whose `cond` differs
    between lanes evaluates the gather for some lanes only - and the others' bits vanish from what the active lanes read (the
    first aligned-window version of gym_emit lost its type planes that way).  The kernels' idiom is: cross-lane reads into
    locals first, then the logic; this scan keeps it that way."""
    import re
    src_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "generalsreinforcementlearning_amd", "csrc")
    pat = re.compile(r"(&&|\|\||\?)\s*!?\s*(at|arm_at|bperm|rdlane_any|b\.gather|b\.gather_mask|gather|gather_mask)\s*\(")
    bad = []
    for name in sorted(os.listdir(src_dir)):
        if name.endswith((".hip", ".hpp")):
            for i, line in enumerate(open(os.path.join(src_dir, name)), 1):
                code = line.split("//")[0]
                if pat.search(code):
                    bad.append(f"{name}:{i}: {line.strip()}")
    assert not bad, "\n".join(bad)
