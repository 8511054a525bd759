"""CPU test (hipcc cross-compiles gfx950 without a GPU): the in-launch hand-off of partial rows to a pair's closing workgroup
(delta_graph_slam_amd/csrc/common.h, "in-launch hand-off") rests on an instruction sequence, not on fences -- so the sequence is
asserted on the compiled kernels.  In each fused kernel, in program order:
  1. the row is stored write-through:            global_store_dwordx2 ... sc1
  2. the storing wave drains:                    s_waitcnt vmcnt(0)          (before the next barrier)
  3. the workgroup meets:                        s_barrier
  4. one lane takes the ticket, agent scope:     global_atomic_add ... sc0   (returns the old value)
  5. the workgroup meets again:                  s_barrier
  6. the closing workgroup reads rows coherently: global_load_dwordx2 ... sc1
and no cache-wide write-back / invalidate (buffer_wbl2 / buffer_inv) is paid for it.  A compiler that reordered or dropped any of
these would fail here instead of giving silently wrong sums; the bit-for-bit tests against the unfused path
(test_fused_launches_equal_launch_pairs_bit_for_bit, test_fused_rounds_equal_separate_solve_launches_bit_for_bit) are the
functional guard on the GPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "delta_graph_slam_amd", "csrc")

KERNELS = {
    "ndt_align.s": ["_ZN3dgs22ndt_derivatives_kernelILi%dELb1ELb0EE" % s for s in (0, 1, 2, 3)]           # <search, fused, scalar>
                   + ["_ZN3dgs18ndt_strict3_kernelILi%dELb1ELb1ELb%dEE" % (s, f) for s in (0, 1, 2, 3) for f in (0, 1)]   # upstream order, item-compacted: <search, fused, with the double pass, fixed slices>
                   + ["_ZN3dgs17ndt_strict_kernelILi2ELb1ELb%dEE" % hd for hd in (0, 1)],                  # upstream order, lane-per-point: <DIRECT7, fused, float kinds / double pass>
    "gicp.s": ["_ZN3dgs21gicp_linearize_kernelILb1EE", "_ZN3dgs22vgicp_linearize_kernelILb1EE"],
}


@pytest.fixture(scope="module")
def asm():
    subprocess.check_call(["make", "-C", CSRC, "isa", "-j2"], stdout=subprocess.DEVNULL)
    return {f: open(os.path.join(CSRC, "build", f)).read() for f in KERNELS}


def _body(text, prefix):
    m = re.search(r"^(%s\w*):" % re.escape(prefix), text, flags=re.M)
    assert m, prefix
    start = m.end()
    return [ln.strip() for ln in text[start:text.index(".Lfunc_end", start)].splitlines()]


def _first(lines, pattern, start=0):
    for i in range(start, len(lines)):
        if re.search(pattern, lines[i]):
            return i
    return -1


@pytest.mark.parametrize("file,prefix", [(f, k) for f, ks in KERNELS.items() for k in ks])
def test_fused_kernels_hand_their_rows_over_in_the_documented_order(asm, file, prefix):
    ln = _body(asm[file], prefix)
    store = _first(ln, r"^global_store_dwordx2 .* sc1\b")
    assert store >= 0, "row store is not write-through (sc1)"
    barrier1 = _first(ln, r"^s_barrier", store)
    drain = _first(ln, r"^s_waitcnt vmcnt\(0\)$", store)
    assert 0 <= drain < barrier1, "the storing wave does not drain its stores before the barrier"
    ticket = _first(ln, r"^global_atomic_add \S+, \S+, \S+, .* sc0\b", barrier1)
    assert ticket > barrier1, "the ticket is not taken behind the barrier"
    assert _first(ln, r"^global_atomic_add", store) == ticket, "another atomic sits between the row store and the ticket"
    barrier2 = _first(ln, r"^s_barrier", ticket)
    load = _first(ln, r"^global_load_dwordx2 .* sc1\b", ticket)
    assert barrier2 > ticket and load > barrier2, "the closing workgroup's row loads are not coherent (sc1) loads behind the second barrier"
    reset = _first(ln, r"^global_store_dword \S+, \S+, .* sc1\b", ticket)
    assert ticket < reset < barrier2, "the ticket is not reset by its taker with an agent-scope store"
    assert not any(re.match(r"buffer_wbl2|buffer_inv", x) for x in ln), "a cache-wide write-back / invalidate crept into the kernel"


def test_unfused_kernels_do_not_pay_for_the_hand_off(asm):
    for file, prefix in (("ndt_align.s", "_ZN3dgs22ndt_derivatives_kernelILi2ELb0ELb0EE"), ("ndt_align.s", "_ZN3dgs18ndt_strict3_kernelILi2ELb0ELb1ELb0EE"), ("ndt_align.s", "_ZN3dgs18ndt_strict3_kernelILi2ELb0ELb1ELb1EE"),
                         ("gicp.s", "_ZN3dgs21gicp_linearize_kernelILb0EE")):
        ln = _body(asm[file], prefix)
        assert not any(re.search(r"\bsc1\b|global_atomic", x) for x in ln), prefix
