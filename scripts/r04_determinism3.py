"""Reproducibility of a bench workload's forward + backward: N repetitions on one batch without an optimizer step, every
parameter gradient and the loss compared bitwise with the first repetition (checksums, so the 7B model fits)."""
import os, sys, collections, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import bench
import touchnet_amd.specs  # noqa: F401
from touchnet_amd.bin.train import Trainer
name = sys.argv[1] if len(sys.argv) > 1 else "qwen2_audio_7b"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
wl = bench.Workload(name, dev, 0, None, None)
tr = Trainer(wl.job, wl.model_config, dev)
data = tr.next_batch(wl.make_batch())
def digest(t):
    v = t.detach().reshape(-1).view(torch.int16 if t.element_size() == 2 else torch.int32).to(torch.int64)
    return int((v * (torch.arange(v.numel(), device=v.device) % 8191 + 1)).sum())
bad, ref = collections.Counter(), None
for it in range(N):
    tr.optimizer.zero_grad()
    loss, _, _ = tr.forward_loss(data)
    loss.backward()
    from touchnet_amd.models.backend import ops as _ops
    torch.cuda.synchronize()
    cur = {"loss": digest(loss.float())}
    cur.update({n: digest(p.grad) for n, p in tr.model.named_parameters() if p.grad is not None})
    if ref is None:
        ref = cur
        print(name, "loss", float(loss), "gradients:", len(cur) - 1, flush=True)
        continue
    for n in cur:
        if cur[n] != ref[n]:
            bad[n] += 1
print(name, "all reproducible over", N, "repetitions" if not bad else f"NOT reproducible: {len(bad)} tensors, e.g. {dict(list(bad.items())[:12])}")
