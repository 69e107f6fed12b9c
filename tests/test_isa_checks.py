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
    assert r.stdout.count("attn_bwd_") >= 7, r.stdout      # two dK/dV (D=128) + one (D=64) + two dQ + two stream dQ kernels
    assert r.stdout.count("attn_fwd_stream") >= 2, r.stdout


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


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_fused_dkdv_accumulators_and_stage_ring(tmp_path):
    """csrc/attn_bwd_fused.hip keeps dK^T / dV^T in the LITERAL registers a[0:127] at one wave per SIMD and runs every
    MFMA as inline asm.  Sound only while hipcc never touches the accumulator file (no spill, no scratch, every
    v_accvgpr_* and v_mfma inside an asm region, < 256 VGPRs of its own), and fast only while the stage loop keeps the
    counted vmcnt(5) wait of the four-slot LDS-DMA ring and no compiler-inserted vmcnt(0)."""
    import re
    out = tmp_path / "fused.s"
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-S",
                        "--cuda-device-only", os.path.join(ROOT, "touchnet_amd", "csrc", "attn_bwd_fused.hip"),
                        "-o", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    text = out.read_text()
    m = re.search(r"^(_ZN2tn24attn_bwd_kv_fused_kernel\S*):(.*?)s_endpgm", text, re.S | re.M)
    assert m
    body = m.group(2)
    in_asm, mine, theirs, mfma_in, mfma_out = False, 0, [], 0, 0
    for line in body.split("\n"):
        t = line.strip()
        if t.startswith(";;#ASMSTART") or t.startswith(";#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND") or t.startswith(";#ASMEND"):
            in_asm = False
        elif t.startswith("v_accvgpr"):
            if in_asm:
                mine += 1
            else:
                theirs.append(t)
        elif t.startswith("v_mfma"):
            mfma_in, mfma_out = mfma_in + in_asm, mfma_out + (not in_asm)
        assert not t.startswith("scratch_"), t
    # 128 zeroing writes + 128 epilogue reads; five loop-trip variants with MFMAs (32 + 32 + 16 + 16 + 16)
    assert mfma_out == 0 and mine == 256 and mfma_in == 112, (mine, mfma_in, mfma_out)
    # hipcc may park a few of its own values in accumulator registers ABOVE the literal block a[0:127] (outside the stage
    # loop: checked below), never inside it
    for t in theirs:
        idx = [int(x) for x in re.findall(r"\ba(\d+)\b", t)]
        assert idx and all(i >= 128 for i in idx), t
    meta = re.search(r"\.agpr_count:\s+(\d+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)"
                     r".*?\.vgpr_spill_count:\s+(\d+)", text, re.S)
    assert meta, "kernel metadata not found"
    agpr, priv, vgpr, spill = map(int, meta.groups())
    assert 128 <= agpr <= 136 and priv == 0 and spill == 0 and vgpr - agpr <= 256, (agpr, priv, vgpr, spill)
    # the stage loop = the innermost loop holding the barriers of the loop trips
    ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    labels = {mm.group(1): i for i, l in enumerate(ins) for mm in [re.match(r"(\.LBB\d+_\d+):", l)] if mm}
    loops = []
    for i, l in enumerate(ins):
        mm = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if mm and labels.get(mm.group(1), len(ins)) < i:
            seg = ins[labels[mm.group(1)]:i + 1]
            if sum(x.startswith("v_mfma") for x in seg) == 112:
                loops.append(seg)
    assert loops, "stage loop not found"
    seg = min(loops, key=len)
    vm = [x for x in seg if x.startswith("s_waitcnt") and "vmcnt" in x]
    assert vm == ["s_waitcnt vmcnt(5)"] * 6, vm          # one counted wait per trip variant, nothing else
    # no SGPR spill traffic and none of hipcc's parked accumulator values inside the loop
    assert not any(x.startswith(("v_readlane", "v_writelane")) for x in seg)
    assert sum(x.startswith("v_accvgpr") for x in seg) == 0
    assert sum(x.startswith("v_pk_") for x in seg) == 0, "packed f32 VALU beside the MFMAs (see the kernel's ds_elem)"


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_product_gemm_kernels_keep_their_stage_loops_clean(tmp_path):
    """Every instantiation of the product GEMM (csrc/gemm.hip: `gemm_kernel` with its epilogue modes — plain, grouped, SwiGLU
    forward / backward, bias gradient, RoPE — and `gemm16_kernel`, the 16x16x32 variant, with the GELU epilogues of round 6)
    compiled for gfx950: no scratch, no
    VGPR spill (a fused epilogue must not push the main loop out of the register file); each kernel issues ONE MFMA shape;
    and its stage loops hold no `s_waitcnt vmcnt` but the hand-counted `vmcnt(4)` of the five-slot LDS-DMA ring (a
    compiler-inserted one in front of an LDS access would drain the ring every stage)."""
    import re
    out = tmp_path / "gemm.s"
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "touchnet_amd", "csrc", "gemm.hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    text = out.read_text()
    kernels = re.findall(r"^(_ZN2tn4gemm(?:11gemm_kernel|13gemm16_kernel)\S*):(.*?)s_endpgm", text, re.S | re.M)
    names = [n for n, _ in kernels]
    # (gemm16: plain x 2 operand modes, SwiGLU forward / backward, RoPE, GELU forward / backward)
    assert sum("gemm16_kernel" in n for n in names) == 7 and sum("11gemm_kernel" in n for n in names) >= 14, names
    meta = text[text.find("amdhsa.kernels"):]
    for ent in meta.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", ent).group(1)
        if "gemm" not in name or "splitk_reduce" in name:
            continue
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", ent).group(1)) == 0, name
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", ent).group(1)) == 0, name
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", ent).group(1)) <= 256, name
    for name, body in kernels:
        ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        m16 = "gemm16_kernel" in name
        shapes = {l.split()[0] for l in ins if l.startswith("v_mfma")}
        assert shapes == ({"v_mfma_f32_16x16x32_bf16"} if m16 else {"v_mfma_f32_32x32x16_bf16"}), (name, shapes)
        per_trip = 64 if m16 else 32
        labels = {m.group(1): i for i, l in enumerate(ins) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
        found = 0
        for i, l in enumerate(ins):
            m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), len(ins)) < i:
                seg = ins[labels[m.group(1)]:i + 1]
                if sum(x.startswith("v_mfma") for x in seg) == per_trip:          # the steady-state stage loop
                    found += 1
                    vm = [x for x in seg if x.startswith("s_waitcnt") and "vmcnt" in x]
                    assert vm == ["s_waitcnt vmcnt(4)"], (name, vm)
        assert found == 1, (name, found)
