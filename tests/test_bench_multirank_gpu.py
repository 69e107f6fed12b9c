"""bench.py's N > 1 code path end to end on ONE MI355X: the driver's 8-GPU launch line, with two
ranks sharing the GPU over gloo (TN_DIST_BACKEND=gloo; RCCL refuses two ranks on one device) — process-group init, mesh
build (dp / cp / tp), the Trainer with its engines, barrier + synchronize fences, the MAX all-reduce of the elapsed time and
rank 0's JSON line.  Small workload ("tiny"): what is under test is the bench's own plumbing, not a number."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(extra, n=2, backend="gloo"):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TN_DIST_BACKEND", None)
    if backend == "gloo":
        env["TN_DIST_BACKEND"] = "gloo"
    env.pop("TN_FORCE_FSDP", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--workload", "tiny", "--no-cpu-baseline", "--no-kernel-rooflines", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]              # rank 0 prints ONE line
    return json.loads(lines[0])


_ONE = {}


def _one_gpu_loss_cached():
    if "loss" not in _ONE:
        _ONE["loss"] = _one_gpu_loss()
    return _ONE["loss"]


@pytest.mark.parametrize("extra,label", [([], "dp2"), (["--dp-engine", "fsdp2"], None), (["--cp", "2"], "cp2"),
                                         (["--tp", "2"], "tp2")])
def test_bench_two_ranks_on_one_gpu(extra, label):
    if extra == ["--dp-engine", "fsdp2"]:
        pytest.skip("FSDP2's DTensor mesh must be a cuda mesh (RCCL): not runnable with two ranks on one GPU")
    line = _launch(extra)
    # context- and tensor-parallel ranks work on ONE batch — the one-GPU run's: the JOB's loss (`loss_per_sample_job`: rank
    # 0's share summed over the data- / context-parallel ranks) is the one-GPU loss, bf16 summation order apart; rank 0's own
    # line entry is its share (half the sequence under cp 2)
    if label in ("cp2", "tp2"):
        one = _one_gpu_loss_cached()
        assert abs(line["loss_per_sample_job"] - one) / abs(one) < 2e-2, (label, line["loss_per_sample_job"], one)
    if label == "cp2":
        assert line["loss_per_sample_last"] < 0.75 * line["loss_per_sample_job"]
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["loss_per_sample_last"] == line["loss_per_sample_last"]
    if label is not None:
        assert label in line["config"]["parallelism"], line["config"]["parallelism"]
    # the line says what it ran on, read from the live process group
    d = line["dist"]
    assert d["initialized"] and d["backend"] == "gloo" and d["world_size"] == 2 and d["ranks_answering_all_reduce"] == 2
    # whole-job tokens: dp ranks each bring B x T, cp / tp peers share one batch
    per_rank = 1 * 512
    dp = 2 if not extra else 1
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - per_rank * dp) / (per_rank * dp) < 0.02


def _one_gpu_loss():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TN_FORCE_FSDP", "TN_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--workload", "tiny", "--no-cpu-baseline", "--no-kernel-rooflines"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][0])
    return line["loss_per_sample_last"]


def _devices():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_devices() < 2, reason="needs two MI355X: one RCCL rank per device (the driver's multi-GPU node)")
def test_bench_two_ranks_over_rccl_on_two_gpus():
    """The same four layouts over REAL RCCL, one rank per device (collected everywhere, runs where two GPUs are visible — the
    driver's scaling node): the line must say backend nccl (= RCCL on ROCm), world size 2 and two ranks answering an
    all-reduce; context- and tensor-parallel ranks work on ONE batch, so their loss is the one-GPU run's (bf16 summation
    order apart); the two data-parallel engines see the same two batches and must agree with each other.  Reference
    layouts: touchnet/models/helper_func.py:134-202 (FSDP2), touchnet/utils/distributed.py:292-315 (CP),
    touchnet/models/llama/parallelize_llama.py:105-196 (TP)."""
    one = _one_gpu_loss()
    got = {}
    for extra in ([], ["--dp-engine", "fsdp2"], ["--cp", "2"], ["--tp", "2"]):
        line = _launch(extra, backend="nccl")
        d = line["dist"]
        assert d["backend"] == "nccl" and d["world_size"] == 2 and d["ranks_answering_all_reduce"] == 2, d
        assert d["devices_visible"] >= 2 and d["rccl_version"], d
        got[" ".join(extra) or "flat"] = line["loss_per_sample_job"]
    for k in ("--cp 2", "--tp 2"):
        assert abs(got[k] - one) / abs(one) < 2e-2, (k, got, one)
    assert abs(got["flat"] - got["--dp-engine fsdp2"]) / abs(got["flat"]) < 2e-2, got


def test_plain_command_line_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher re-executes itself under
    torch.distributed.run (bench.self_launch) and rank 0 prints the one JSON line — the route the driver's
    multi-GPU scaling run takes if it starts the plain command."""
    env = dict(os.environ, TN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TN_FORCE_FSDP"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "tiny", "--no-cpu-baseline", "--no-kernel-rooflines"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and "dp2" in line["config"]["parallelism"]


def test_more_ranks_than_gpus_is_refused_without_the_gloo_switch():
    """RCCL needs one device per rank: the plain command line says so instead of hanging in the rendezvous"""
    import torch
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TN_DIST_BACKEND"):
        env.pop(k, None)
    n = torch.cuda.device_count() + 1          # (>= 2 on a GPU box)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "tiny"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stdout + r.stderr)
