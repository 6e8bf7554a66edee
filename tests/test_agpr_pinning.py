"""CPU (needs hipcc, no GPU): the MFMA kernels that pin their accumulator tiles to AGPRs BY NAME inside inline asm
(gram_mfma.hip, lincomb_mfma.hip via agpr_tiles.inc) are only correct if the compiler never touches an AGPR of its own
between the first and the last of those statements — the clobber lists do not stop it from parking a temporary in a named
tile between two statements that use it (design rules; this is how two panel-update variants went wrong in
round 2).  The disassembly is checked: a compiler-generated v_accvgpr_write may only target an AGPR that no LATER
named-tile statement of the kernel uses (spills above the named range, or into a tile the epilogue has already read)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "gcge_amd", "csrc", "hip")
AGPR = re.compile(r"(?<![\w.])a\d+\b|(?<![\w.])a\[\d+:\d+\]")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
@pytest.mark.parametrize("src", ["gram_mfma.hip", "lincomb_mfma.hip"])
def test_named_agpr_tiles_are_not_touched_by_the_compiler(tmp_path, src):
    out = str(tmp_path / (src + ".s"))
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + HIP, "-S",
                    "--cuda-device-only", os.path.join(HIP, src), "-o", out], check=True, capture_output=True)
    def regs(code):
        out = set()
        for m in re.finditer(r"(?<![\w.])a\[(\d+):(\d+)\]", code):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"(?<![\w.])a(\d+)\b", code):
            out.add(int(m.group(1)))
        return out

    kernels = {}          # name -> (list of AGPR sets per inline-asm block, list of (line, blocks seen, code) outside asm)
    kernel, inasm = None, False
    for ln, line in enumerate(open(out), 1):
        m = re.match(r"^(_Z\w*kernel\w*):", line)
        if m:
            kernel, inasm = m.group(1), False
            kernels[kernel] = ([], [])
            continue
        if kernel is None:
            continue
        if line.startswith(".Lfunc_end"):
            kernel = None
            continue
        blocks, outside = kernels[kernel]
        if "ASMSTART" in line:
            inasm = True; blocks.append(set())
            continue
        if "ASMEND" in line:
            inasm = False
            continue
        code = line.split(";")[0]
        if not code.strip() or code.lstrip().startswith("."):
            continue
        r = regs(code)
        if r:
            if inasm:
                blocks[-1].update(r)
            elif code.split()[0].startswith("v_accvgpr_write"):      # the compiler parks a value in an AGPR
                outside.append((ln, len(blocks), code.strip(), regs(code.split(",")[0])))
    checked, report = 0, []
    for name, (blocks, outside) in kernels.items():
        named = set().union(*blocks) if blocks else set()
        if not named:
            continue
        checked += 1
        bad = []
        for ln, seen, code, dst in outside:
            later = set().union(*blocks[seen:]) if seen < len(blocks) else set()
            if seen > 0 and dst & later:          # a named tile register that a LATER named-tile statement still uses
                bad.append((ln, code))
        if bad:
            report.append("%s: the compiler writes %d times into named accumulator registers that are still live, e.g. %r" % (name, len(bad), bad[0]))
    assert checked > 0, "no kernel with named AGPR tiles found in " + src
    assert not report, "\n".join(report)
