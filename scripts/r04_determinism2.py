"""Which gradient is not reproducible?  One model, one batch, forward + backward N times without an optimizer step; every
parameter gradient (and the loss) is compared bitwise with the first repetition's."""
import os, sys, collections, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import touchnet_amd.specs  # noqa
from touchnet_amd.bin.train import TrainConfig, Trainer
from touchnet_amd.data.synthetic import text_batch
from touchnet_amd.models.llama import DecoderConfig
CFG = dict(model_type="llama", hidden_size=2560, intermediate_size=2560, num_attention_heads=20, num_hidden_layers=2,
           num_key_value_heads=20, head_dim=128, vocab_size=1024, tie_word_embeddings=False, rope_theta=500000.0,
           initializer_range=0.02)
cfg = DecoderConfig.from_dict(CFG)
job = dict(training_model_name="llama_mi355", training_enable_fused_ce=True, lr_scheduler_warmup_steps=0, lr_scheduler_lr=1e-3)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
tr = Trainer(TrainConfig(**job), cfg, torch.device("cuda", 0))
bad = collections.Counter()
for s in range(2):
    data = tr.next_batch(text_batch(1024, 4, 512, seed=s, max_len=90))
    ref = None
    for it in range(N):
        tr.optimizer.zero_grad()
        loss, _, _ = tr.forward_loss(data)
        loss.backward()
        torch.cuda.synchronize()
        cur = {"loss": loss.detach().clone()}
        cur.update({n: p.grad.detach().clone() for n, p in tr.model.named_parameters()})
        if ref is None:
            ref = cur
            continue
        for n in cur:
            if not torch.equal(cur[n], ref[n]):
                d = (cur[n].float() - ref[n].float()).abs()
                bad[n] += 1
                if bad[n] <= 2:
                    print(f"  batch {s} rep {it}: {n} differs: max {float(d.max()):.3e} at {int((d > 0).sum())} of {d.numel()} elements")
print(os.environ.get("TAG", ""), "all reproducible" if not bad else dict(bad))
