"""Where does the 3.7 % error of the Kimi fixture's k_proj-bias gradient sit?  Runs the device model on the
reference-run fixture and prints, for the worst k_proj bias, error and reference magnitude per rotary frequency index."""
import ast
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from touchnet_amd.loss.cross_entropy import cross_entropy_loss  # noqa: E402
from touchnet_amd.models.kimi_audio import KimiAudioConfig, KimiAudioPackedForCausalLM  # noqa: E402
import test_reference_fixtures_gpu as T  # noqa: E402

g = np.load(os.path.join(os.path.dirname(T.__file__), "golden", "kimi_decoder_dev.npz"), allow_pickle=True)
kw = ast.literal_eval(str(g["config_json"]))
kw["head_dim"] = kw["hidden_size"] // kw["num_attention_heads"]
m = KimiAudioPackedForCausalLM(KimiAudioConfig(**{k: v for k, v in kw.items() if k != "initializer_range"}))
m.load_state_dict(T._state(g), strict=True)
m = m.to("cuda").to(torch.bfloat16)
d = {k: v.to("cuda") for k, v in T._batch(g).items()}
out = m(text_input_ids=d["text_input_ids"], audio_input_ids=d["audio_input_ids"], attention_mask=d["attention_mask"],
        position_ids=d["position_ids"], compute_audio_logits=True)
ps, _ = cross_entropy_loss(out.logits, d["labels"], d["sentence_lens"], 4)
ps.backward()
D = kw["head_dim"]
rows = []
for n, p in m.named_parameters():
    if n.endswith(("k_proj.bias", "q_proj.bias", "v_proj.bias")) and "grad/" + n in g.files:
        r = g["grad/" + n].astype(np.float32)
        e = np.abs(p.grad.float().cpu().numpy() - r)
        rows.append((e.max() / np.abs(r).max(), n, e, r))
rows.sort(key=lambda t: -t[0])
for rel, n, e, r in rows[:6]:
    i = int(e.argmax())
    print(f"{n:45s} worst {rel * 100:5.2f} % of scale at element {i} (dim {i % D} of its head, rotary index {i % (D // 2)} of {D // 2}); "
          f"|ref| there {abs(r[i]):.2e}, scale {np.abs(r).max():.2e}")
rel, n, e, r = next(t for t in rows if "k_proj" in t[1])
fr = np.arange(r.size) % (D // 2)
print("k_proj bias, per rotary-frequency quartile (0 = fastest rotation): mean |ref|, mean |err|")
for q in range(4):
    sel = (fr * 4 // (D // 2)) == q
    print(f"  quartile {q}: |ref| {np.abs(r[sel]).mean():.2e}  |err| {e[sel].mean():.2e}")
