"""CPU-side check of the emitted gfx950 ISA of pr_cold_tile_kernel (hipcc cross-compiles without a GPU).

The kernel's prefetch is inline asm with hand-counted `s_waitcnt vmcnt(N)` (gmx_pr_cold.hip: prc_set_load /
PRC_SET_WAIT); its correctness rests on what the compiler emits between a load and its wait.  tools/isa_check.py proves
on the control-flow graph of the assembly that every wait is sufficient on every path, that no register of a prefetch
set is touched while its load may be in flight, and that nothing spills."""
import os
import shutil
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "green-marl_amd", "tools"))
SRC = os.path.join(ROOT, "green-marl_amd", "csrc", "gmx_pr_cold.hip")

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")


def test_tile_kernel_waits_are_sufficient_and_nothing_spills():
    import isa_check
    res = isa_check.check_file(SRC)
    assert len(res) == 6, list(res)                       # fp32 and fp64: class forms (two tile sizes) and generic forms
    for name, (errors, stats) in res.items():
        assert not errors, (name, errors[:5])
        assert stats["scratch"] == 0 and 0 < stats["vgprs"] <= 128, stats
        assert stats["asm_loads"] >= 4, stats
        assert sum(stats["waits"].values()) >= 4, stats


def test_checker_refuses_a_wait_that_is_one_too_high(tmp_path):
    """The checker is not vacuous: the same source with one wait raised from 40 to 41 outstanding operations (what a
    dropped store would amount to) must be refused."""
    import isa_check
    text = open(SRC).read()
    assert "PRC_SET_WAIT(B, 40);" in text
    bad = tmp_path / "gmx_pr_cold_bad.hip"
    bad.write_text(text.replace("PRC_SET_WAIT(B, 40);", "PRC_SET_WAIT(B, 41);", 1))
    shutil.copy(os.path.join(os.path.dirname(SRC), "gmx_internal.h"), tmp_path / "gmx_internal.h")
    res = isa_check.check_file(str(bad))
    assert any(errors for errors, _ in res.values())
