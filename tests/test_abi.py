"""The C-ABI library builds for gfx950, loads on a host without a GPU, and exports every symbol that
include/touchnet_amd.h declares; the ctypes stub covers exactly that set (no compute calls here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "touchnet_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tn_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_header_symbols():
    from touchnet_amd import build
    lib_path = build.build()
    assert os.path.exists(lib_path)
    from touchnet_amd import _C
    lib = _C.lib()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/touchnet_amd.h but not exported"
    assert set(_C.PROTOTYPES) == set(syms), set(_C.PROTOTYPES) ^ set(syms)
    assert "gfx950" in _C.version()


def test_every_entry_point_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "touchnet_amd.h")).read()
    for needle in ("touchnet/loss/cross_entropy.py", "touchnet/data/functions.py", "flex_attention.py",
                   "touchnet/utils/optimizer.py", "modeling_llama.py"):
        assert needle in text


def test_product_never_imports_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "touchnet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_gemm_tuning_replay_env(monkeypatch, tmp_path):
    """gemm_tuning.enable(): replay mode by default (tuning OFF, one results file per device ordinal), untouched when
    the user configured TunableOp, opt-in tuning of new shapes into a caller-chosen directory."""
    import os

    from touchnet_amd.utils import gemm_tuning
    for k in [k for k in os.environ if k.startswith("PYTORCH_TUNABLEOP") or k.startswith("TN_TUNE")]:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.delenv("TN_RECORD_UNTUNED", raising=False)
    assert os.path.exists(gemm_tuning.RESULTS)
    lines = open(gemm_tuning.RESULTS).read().splitlines()
    keys = [tuple(ln.split(",")[:2]) for ln in lines if not ln.startswith("Validator")]
    assert len(keys) == len(set(keys)), "duplicate GEMM entries in the replay table"
    assert gemm_tuning.enable() is True
    assert os.environ["PYTORCH_TUNABLEOP_ENABLED"] == "1" and os.environ["PYTORCH_TUNABLEOP_TUNING"] == "0"
    d = os.path.dirname(os.environ["PYTORCH_TUNABLEOP_FILENAME"])
    assert all(os.path.exists(os.path.join(d, f"results{i}.csv")) for i in range(8))
    assert gemm_tuning.enable() is False                       # already configured -> hands off
    for k in [k for k in os.environ if k.startswith("PYTORCH_TUNABLEOP")]:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TN_TUNE_NEW", "1")
    monkeypatch.setenv("TN_TUNE_DIR", str(tmp_path / "tune"))
    assert gemm_tuning.enable() is True and os.environ["PYTORCH_TUNABLEOP_TUNING"] == "1"
    assert os.environ["PYTORCH_TUNABLEOP_FILENAME"].startswith(str(tmp_path / "tune"))


def test_round5_gemm_entry_points_refuse_malformed_arguments_before_any_launch():
    """The fused-epilogue / grouped GEMM entry points (include/touchnet_amd.h) check shapes, strides and alignment on the host
    and return TN_EINVAL (-22) without touching the device — so the checks can be exercised on a box with no GPU.  (Valid calls
    are the `-m gpu` tests' business: tests/test_gemm_fused_gpu.py.)"""
    import ctypes as C
    from touchnet_amd import _C
    lib = _C.lib()
    EINVAL = -22
    raw = (C.c_char * 8192)()
    p = (C.addressof(raw) + 255) // 256 * 256            # a 256-byte-aligned host address: never dereferenced by a refused call
    odd = p + 2                                          # not 16-byte aligned
    H, I, M = 4096, 11008, 256
    # SwiGLU forward: K not a multiple of 64, an output stride shorter than a row, a misaligned operand
    ok = dict(M=M, I=I, K=H, ldx=H, ldw=H, ldc=I)
    def swiglu_fwd(x=p, **kw):
        a = {**ok, **kw}
        return lib.tn_gemm_bf16_swiglu_fwd(x, p, p, p, p, p, a["M"], a["I"], a["K"], a["ldx"], a["ldw"], a["ldc"], None)
    assert swiglu_fwd(K=H + 32) == EINVAL
    assert swiglu_fwd(ldc=I - 8) == EINVAL
    assert swiglu_fwd(M=0) == EINVAL
    assert swiglu_fwd(x=odd) == EINVAL
    assert swiglu_fwd(ldx=H - 64) == EINVAL and swiglu_fwd(ldw=H - 64) == EINVAL     # a row pitch below the contraction
    # SwiGLU backward
    assert lib.tn_gemm_bf16_swiglu_bwd(p, p, p, p, p, p, M, I, H + 8, I, I, I, None) == EINVAL
    assert lib.tn_gemm_bf16_swiglu_bwd(p, p, p, p, p, p, M, I, H, I, I, I - 8, None) == EINVAL
    assert lib.tn_gemm_bf16_swiglu_bwd(odd, p, p, p, p, p, M, I, H, I, I, I, None) == EINVAL
    assert lib.tn_gemm_bf16_swiglu_bwd(p, p, p, p, p, p, M, I, H, H - 64, I, I, None) == EINVAL      # dY pitch below H
    # RoPE epilogue: head_dim other than 64 / 128, N not a whole number of heads, no tables
    assert lib.tn_gemm_bf16_rope(p, p, None, p, p, p, M, 4096, H, H, H, 4096, 96, None) == EINVAL
    assert lib.tn_gemm_bf16_rope(p, p, None, p, p, p, M, 4096 + 128, H, H, H, 4096 + 128, 128, None) == EINVAL
    assert lib.tn_gemm_bf16_rope(p, p, None, None, None, p, M, 4096, H, H, H, 4096, 128, None) == EINVAL
    assert lib.tn_gemm_bf16_rope(p, p, None, p, p, p, M, 4096, H, H - 64, H, 4096, 128, None) == EINVAL
    assert lib.tn_gemm_bf16_rope(p, p, None, p, p, p, M, 4096, H, H, H - 64, 4096, 128, None) == EINVAL
    # weight gradient + bias gradient without a bias buffer
    assert lib.tn_gemm_bf16_wgrad_bias(p, p, I, H, M, p, None, I, H, H, 0, 0, 1, None, 0, None) == EINVAL
    # grouped launch: no group, too many groups, a mode other than the weight-gradient one
    arr_p, arr_ll, arr_i = (C.c_void_p * 1)(p), (C.c_longlong * 1)(I), (C.c_int * 1)(M)
    Ns, Ms, ldc = (C.c_int * 1)(H), (C.c_int * 1)(I), (C.c_longlong * 1)(H)
    def grouped(ngrp=1, a_kmaj=1, b_kmaj=1):
        return lib.tn_gemm_bf16_grouped(arr_p, arr_p, arr_ll, (C.c_longlong * 1)(H), arr_i, arr_p, ldc, Ms, Ns, ngrp, a_kmaj,
                                        b_kmaj, 0, 0, None, 0, None)
    assert grouped(ngrp=0) == EINVAL
    assert grouped(ngrp=1000) == EINVAL
    assert grouped(a_kmaj=0) == EINVAL


def test_attn_bwd_rope_refuses_missing_or_misaligned_tables_before_any_launch():
    """tn_attn_bwd_rope (round 6: the rotary embedding's backward inside the attention backward) checks its tables and the head
    geometry on the host: TN_EINVAL without touching the device.  (Valid calls: tests/test_kernels_gpu.py, bit-identity with
    tn_attn_bwd + tn_rope_apply.)"""
    import ctypes as C
    from touchnet_amd import _C
    lib = _C.lib()
    EINVAL = -22
    raw = (C.c_char * 4096)()
    p = (C.addressof(raw) + 255) // 256 * 256
    call = lambda cos, sin, D=128, Nh=4, Nkv=4: lib.tn_attn_bwd_rope(p, p, p, p, p, p, p, p, p, p, p, p, 1, 256, Nh, Nkv, D, 0.1,
                                                                     cos, sin, None)
    assert call(None, p) == EINVAL and call(p, None) == EINVAL
    assert call(p + 2, p) == EINVAL and call(p, p + 8) == EINVAL          # 16-byte loads of the table rows
    assert call(p, p, D=96) == EINVAL
    assert call(p, p, Nh=6, Nkv=4) == EINVAL                              # query heads not a multiple of kv heads
