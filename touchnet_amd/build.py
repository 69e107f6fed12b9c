"""Build libtouchnet_amd.so (all HIP kernels + the C ABI) for gfx950 with hipcc.

    python -m touchnet_amd.build            # incremental
    python -m touchnet_amd.build --force

hipcc cross-compiles without a GPU; the .so is written IN-TREE (touchnet_amd/_lib/) so that it
travels to the GPU box with the repo snapshot.  No CUDA / hipify / multi-arch paths: gfx950 only.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_lib")
LIB = os.path.join(OUT, "libtouchnet_amd.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"]


# per-file extras.  -fno-honor-nans: lets fmaxf chains over MFMA outputs become v_max3_f32 without a canonicalising
# v_max per element (the forward softmax's row max); these kernels produce and consume no NaNs (masked scores are
# -inf, fully masked rows are handled explicitly).
EXTRA_FLAGS = {"attn_fwd.hip": ["-fno-honor-nans"], "attn_fwd_pp.hip": ["-fno-honor-nans"],
               "attn_fwd_stream.hip": ["-fno-honor-nans"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: touchnet_amd needs the ROCm toolchain to build its kernels")
    return exe


def _digest(paths) -> str:
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OUT, exist_ok=True)
    srcs = sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    stamp = os.path.join(OUT, "build.stamp")
    want = _digest(srcs + headers)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return LIB
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OUT, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(want)
    if verbose:
        print(f"[touchnet_amd.build] built {LIB} from {len(srcs)} sources", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
