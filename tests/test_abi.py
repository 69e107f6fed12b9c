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
