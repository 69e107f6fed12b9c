"""Compile-time (CPU, hipcc cross-compiles gfx950) checks on the generated ISA of kernels whose speed depends on a
compiler behaviour that the source cannot express."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_attention_backward_stage_ring_is_not_serialised_by_alias_waits():
    """The LDS-DMA stage ring of the dK/dV and dQ kernels relies on counted vmcnt waits.  hipcc inserts a full
    `s_waitcnt vmcnt(0)` in front of LDS reads without alias metadata while a DMA is pending (attn_common.h, i32x4_t;
    DESIGN.md 5.2): the check fails if such a wait appears in front of a row-operand read inside the stage loops."""
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "check_dma_waits.sh")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("attn_bwd_") >= 5, r.stdout      # two dK/dV (D=128) + one (D=64) + two dQ kernels seen
