"""Tune (PyTorch TunableOp) the weight-gradient GEMM shapes functional._LinearGroup introduced and print the
result lines to merge into touchnet_amd/tuning/tunableop_gfx950.csv.  Run on an MI355X:
    python scripts/tune_new_gemms.py > gpurun_out/tune/new.csv"""
import os
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touchnet_amd.utils import gemm_tuning  # noqa: E402

d = tempfile.mkdtemp(prefix="tn_tune_")
shutil.copy(gemm_tuning.RESULTS, os.path.join(d, "results0.csv"))
os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1",
                  PYTORCH_TUNABLEOP_FILENAME=os.path.join(d, "results.csv"),
                  PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="30", PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="5")
import torch  # noqa: E402

bf, dev = torch.bfloat16, "cuda"
M = 16384
import time  # noqa: E402

SHAPES = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(4096, 4096), (12288, 4096), (22016, 4096)]
for N, K in SHAPES:
    dyt = torch.randn(N, M, dtype=bf, device=dev)
    xt = torch.randn(K, M, dtype=bf, device=dev)
    torch.mm(dyt, xt.t())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        torch.mm(dyt, xt.t())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"# tuned dW[{N},{K}] over M={M}: {ms:.3f} ms sustained = {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s", file=sys.stderr,
          flush=True)
import torch.cuda.tunable as tunable  # noqa: E402

old = set(open(gemm_tuning.RESULTS).read().splitlines())
for r in tunable.get_results():                      # (op signature, parameter signature, solution, time ms)
    line = ",".join(str(x) for x in r)
    if line not in old:
        print(line)
