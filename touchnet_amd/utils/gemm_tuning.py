"""Library-GEMM algorithm selection for the shapes of the packed training step.

The decoder/tower projections are plain library GEMMs (hipBLASLt / rocBLAS through `F.linear`); which
solution the library picks per shape matters on MI355X (its default heuristic takes a K-tile-32 kernel for the
16384 x 4096 x 4096 forward, the tuned pick is 40 % faster in isolation).  `touchnet_amd/tuning/
tunableop_gfx950.csv` holds PyTorch-TunableOp results recorded on an MI355X with this image
(`PYTORCH_TUNABLEOP_TUNING=1 python bench.py`, ~20 min); `enable()` replays them with tuning OFF, so start-up
cost is nil and unknown shapes fall back to the library default.  Measured: 931 -> 912 ms/step on
Qwen2-Audio-7B (sustained GEMM rate is power-limited, so the in-situ gain is far below the isolated one).
"""
from __future__ import annotations

import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
RESULTS = os.path.join(os.path.dirname(_HERE), "tuning", "tunableop_gfx950.csv")


def enable(max_devices: int = 8) -> bool:
    """Must run before the first GEMM of the process.  Returns False (and changes nothing) when the results
    file is absent or TunableOp was already configured by the user."""
    if not os.path.exists(RESULTS) or "PYTORCH_TUNABLEOP_ENABLED" in os.environ:
        return False
    d = tempfile.mkdtemp(prefix="tn_tunableop_")
    for i in range(max_devices):                      # TunableOp appends the device ordinal to the file stem
        shutil.copy(RESULTS, os.path.join(d, f"results{i}.csv"))
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    # TN_TUNE_NEW=1: tune the shapes of this run that have no entry yet (entries already in RESULTS are kept as they
    # are); the merged table is written to $PYTORCH_TUNABLEOP_FILENAME's directory at exit -> copy the new lines over
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1" if os.environ.get("TN_TUNE_NEW") == "1" else "0"
    if os.environ.get("TN_TUNE_NEW") == "1":
        os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "30")
        os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "5")
        d = os.environ.get("TN_TUNE_DIR") or d
        os.makedirs(d, exist_ok=True)
        for i in range(max_devices):
            shutil.copy(RESULTS, os.path.join(d, f"results{i}.csv"))
    # TN_RECORD_UNTUNED=<file stem>: list the GEMMs of a run that have no entry yet (then tune them offline with
    # torch.cuda.tunable.tune_gemm_in_file / scripts/tune_new_gemms.py and append the lines to RESULTS)
    rec = os.environ.get("TN_RECORD_UNTUNED")
    os.environ["PYTORCH_TUNABLEOP_RECORD_UNTUNED"] = "1" if rec else "0"
    if rec:
        os.environ["PYTORCH_TUNABLEOP_UNTUNED_FILENAME"] = rec
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(d, "results.csv")
    return True
