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


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_gemm_accumulators_stay_out_of_the_compilers_hands(tmp_path):
    """The four-wave GEMM keeps its 256 accumulators in the LITERAL registers a[0:255] (asm MFMAs that list them as
    clobbers, csrc/gemm.hip `acc_mfma`).  That is only sound while hipcc itself never touches the accumulator file:
    no spill, no scratch, every v_accvgpr_* inside an inline-asm region; and the stage loops must hold nothing but the
    hand-placed waits (a compiler-inserted `s_waitcnt vmcnt(N)` in front of an LDS read drains the DMA ring)."""
    import re
    out = tmp_path / "gemm.s"
    # (the four-wave geometry is variant 2003 of the source; the product library is built with the eight-wave default)
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-S",
                        "-DTN_GEMM_DEFAULT_VARIANT=2003",
                        "--cuda-device-only", os.path.join(ROOT, "touchnet_amd", "csrc", "gemm.hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    text = out.read_text()
    kernels = re.findall(r"^(_ZN2tn4gemm12gemm4_kernel\S*):(.*?)s_endpgm", text, re.S | re.M)
    assert len(kernels) == 6                       # 3 operand-mode pairs x with / without the transposed copy
    for name, body in kernels:
        in_asm, mine, theirs, mfma_outside = False, 0, 0, 0
        for line in body.split("\n"):
            t = line.strip()
            if t.startswith(";;#ASMSTART") or t.startswith(";#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND") or t.startswith(";#ASMEND"):
                in_asm = False
            elif t.startswith("v_accvgpr"):
                mine, theirs = mine + in_asm, theirs + (not in_asm)
            elif t.startswith("v_mfma") and not in_asm:
                mfma_outside += 1
            assert not t.startswith("scratch_"), (name, t)
        assert theirs == 0 and mfma_outside == 0 and mine == 3 * 256, (name, mine, theirs, mfma_outside)
    for field in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
        for m in re.finditer(rf"\.name:\s+_ZN2tn4gemm12gemm4_kernel.*?\.{field}:\s+(\d+)", text, re.S):
            pass
    meta = re.findall(r"\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", text, re.S)
    assert meta and all(int(a) == 0 and int(b) == 0 for a, b in meta), meta
    # stage loops: the innermost loops with 64 MFMAs hold exactly four lgkmcnt(0) waits + one vmcnt(8), nothing else
    for name, body in kernels:
        ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        labels = {m.group(1): i for i, l in enumerate(ins) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
        found = 0
        for i, l in enumerate(ins):
            m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), len(ins)) < i:
                seg = ins[labels[m.group(1)]:i + 1]
                if sum(x.startswith("v_mfma") for x in seg) == 64:
                    found += 1
                    waits = sorted(x for x in seg if x.startswith("s_waitcnt"))
                    assert waits == ["s_waitcnt lgkmcnt(0)"] * 4 + ["s_waitcnt vmcnt(8)"], (name, waits)
        assert found == 4, (name, found)            # one stage loop per wave
