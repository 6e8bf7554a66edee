"""CPU (needs hipcc, no GPU): the third form of the plane sweep (spmm_star.hip: spmm_star3_kernel) mixes LDS-DMA pieces issued from
inline asm with loads and stores hipcc manages, all on ONE in-order vmcnt queue; its hand-written `s_waitcnt vmcnt(N)` before the
strips of plane z are copied assumes a MINIMUM number of operations issued since those strips were requested two steps earlier
(DESIGN.md §3): per step three DMA pieces, the request of the own point, the diagonal (and the clean flag with column sums), and at
most one result store.  The disassembly is checked step by step, so that a compiler that reorders, duplicates or drops one of
these — or spills a register to scratch, which is a vector-memory operation it would wait for at once — fails here and not as a
wrong product on the GPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "gcge_amd", "csrc", "hip")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_hand_counted_waits_of_the_dma_sweep_match_the_disassembly(tmp_path):
    out = str(tmp_path / "spmm_star.s")
    subprocess.run(["hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + HIP, "-S",
                    "--cuda-device-only", os.path.join(HIP, "spmm_star.hip"), "-o", out], check=True, capture_output=True)
    check_assembly(out)


def check_assembly(out):
    """The checks on a device assembly file of spmm_star.hip (`make -C gcge_amd/csrc check-sweep` runs them on the file compiled with the
    Makefile's own flags: python3 tests/test_star_dma_counts.py build/spmm_star.s)."""
    kernels, name = {}, None
    for line in open(out):
        m = re.match(r"^(_ZN4gcge17spmm_star3_kernelILb([01])ELb([01])E\w*):", line)
        mm = re.match(r"^(_ZN4gcge18spmm_star3m_kernelILb([01])E\w*):", line)
        if m:
            name = m.group(1); kernels[name] = {"dot": m.group(2) == "1", "masked": False, "code": []}
            continue
        if mm:
            name = mm.group(1); kernels[name] = {"dot": mm.group(2) == "1", "masked": True, "code": []}
            continue
        if name and line.startswith(".Lfunc_end"):
            name = None
        elif name:
            code = line.split(";")[0].strip()
            if code and not code.startswith("."):
                kernels[name]["code"].append(code)
    # four product kernels (DOT x SLAB) + the timing probe of round 5 (third template flag: own points requested at the strips' lead,
    # tools/star_lead_probe.py) — the probe issues the same operations per step, so it is held to the same counts
    # + the two kernels of the third form on MASKED grids (round 5: spmm_star3m_kernel<DOT>): the same protocol, the clean flag requested
    # with and without DOT (six operations per step either way), the row lookups SCALAR loads (no entry in the vmcnt queue)
    assert len(kernels) == 7 and sum(n.startswith("_ZN4gcge17spmm_star3_kernelILb0ELb0ELb1E") for n in kernels) == 1, list(kernels)
    assert sum(k["masked"] for k in kernels.values()) == 2
    for name, k in kernels.items():
        want = 9 if (k["dot"] or k["masked"]) else 7
        code = k["code"]
        # the steady-state steps: from one hand-written wait to the next
        marks = [i for i, c in enumerate(code) if c == "s_waitcnt vmcnt(%d)" % want]
        assert len(marks) >= 14, (name, len(marks))
        # no scratch access inside the sweep (hipcc would wait for it at once and drain the prefetches); the masked kernel with column
        # sums keeps ONE value in scratch from the prologue to the epilogue (128 of 128 registers): outside the steps, harmless
        inside = [c for c in code[marks[0]:marks[-1]] if "scratch_" in c]
        assert not inside, (name, "a register is spilled to scratch inside the sweep", inside)
        if not k["masked"]:
            assert not any("scratch_" in c for c in code), name + ": a register was spilled to scratch"
        else:
            assert sum("scratch_" in c for c in code) <= 2, name
            assert not any(re.match(r"global_load_dword\b", c) for c in code[marks[0]:marks[-1]]), name + ": a row lookup became a vector load"
        for a, b in zip(marks, marks[1:]):
            step = code[a:b]
            dma = sum(c.startswith("global_load_lds_dwordx4") for c in step)
            loads = sum(bool(re.match(r"global_load_(dwordx4|dwordx2|ubyte|dword)\b", c)) for c in step)
            stores = sum(c.startswith("global_store") for c in step)
            if dma == 0:        # (the stretch between the last step of the unrolled body and the loop head / epilogue)
                continue
            assert dma == 3, (name, "pieces per step", dma)
            assert loads == (3 if (k["dot"] or k["masked"]) else 2), (name, "compiler loads per step", loads, [c for c in step if c.startswith("global_load")])
            # order inside a step: the three pieces first, then the compiler's loads — the count behind a wait is only a
            # lower bound on what has been issued since the awaited pieces if nothing of a later step slips in front of them
            idx_dma = [i for i, c in enumerate(step) if c.startswith("global_load_lds_dwordx4")]
            idx_ld = [i for i, c in enumerate(step) if re.match(r"global_load_(dwordx4|dwordx2|ubyte|dword)\b", c)]
            assert max(idx_dma) < min(idx_ld), (name, "a compiler load was moved in front of the DMA pieces of its step")
            assert stores <= 1, (name, "stores per step", stores)
            # two steps' worth of operations behind the strips being waited for: 2 x (3 pieces + loads) - 3 = loads + 3 + loads
            assert 2 * loads + 3 == want, (name, loads, want)
        assert code.count("s_waitcnt vmcnt(0)") >= 2, name          # prologue and the drain before the block releases its LDS


if __name__ == "__main__":
    import sys
    check_assembly(sys.argv[1])
    print("plane sweep: hand-counted waits match the disassembly of %s" % sys.argv[1])
