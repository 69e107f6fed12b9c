"""Run-to-run determinism of the plain Trainer on the 2560-wide two-layer model of tests/test_parallel_gpu.py (split-K
GEMMs, D = 128 attention): the same five steps three times in one process; prints every run's losses + last norm."""
import os, sys, torch
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
batches = [text_batch(1024, 4, 512, seed=s, max_len=90) for s in range(4)]
def run():
    tr = Trainer(TrainConfig(**job), cfg, torch.device("cuda", 0))
    out = []
    for b in batches + batches[:1]:
        r = tr.train_step(tr.next_batch(b))
        out.append((round(float(r["loss_per_sample"]), 6), round(float(r["grad_norm"]), 5)))
    return out
runs = [run() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4)]
same = all(r == runs[0] for r in runs)
print(os.environ.get("TAG", ""), "DETERMINISTIC" if same else "DIFFERS")
if not same:
    for r in runs:
        print("   ", r)
