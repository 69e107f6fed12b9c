"""Generate the golden fixtures in this directory BY RUNNING THE REFERENCE.

Run once in the build container (where /root/reference is mounted read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's own functions (recipe: SURVEY.md §8c, _ref_import.py) and
the third-party executors it drives (transformers Llama / Qwen2AudioEncoder, flex
mask builder), feeds them seeded synthetic inputs and stores inputs + outputs as
small .npz files.  The fixtures are DATA ONLY; no reference source is stored.
The GPU box never runs this script (no /root/reference there).
"""
import os
import sys
import types
import wave

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402

R.install()

from touchnet.data import functions as ref_fn  # noqa: E402
from touchnet.loss.cross_entropy import cross_entropy_loss as ref_ce  # noqa: E402
from touchnet.models.llama.processing_llama import batch_text as ref_batch_text  # noqa: E402
from touchnet.models.touch_audio.processing_touch_audio import \
    batch_pairaudio_pairtext_packed as ref_batch_asr  # noqa: E402

TOK = types.SimpleNamespace(bos=1, eos=2, pad=0)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path)} bytes")


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


# ------------------------------------------------------------------ packers
def text_cases():
    rng = np.random.RandomState(2025)
    cases = {}
    # (name, B, T, drop_last, lengths)
    cases["overflow"] = (3, 16, False, [int(x) for x in rng.randint(1, 9, size=40)])
    cases["exactfit"] = (2, 8, False, [3, 3, 7, 1, 5, 7, 7])          # 4+4 | 8 | 2+6 | 8 8
    cases["droplast"] = (2, 12, True, [int(x) for x in rng.randint(1, 11, size=17)])
    cases["single"] = (4, 64, False, [int(x) for x in rng.randint(1, 13, size=9)])
    out = {}
    for name, (B, T, drop, lens) in cases.items():
        sents = [[int(v) for v in rng.randint(3, 16, size=n)] for n in lens]
        cfg = types.SimpleNamespace(dataset_batchsize=B, dataset_text_seqlen=T,
                                    dataloader_drop_last_batch=drop)
        batches = list(ref_batch_text(({"input_ids": s} for s in sents), cfg, TOK))
        out[f"{name}/meta"] = np.array([B, T, int(drop), len(batches)])
        out[f"{name}/lens"] = np.array(lens)
        out[f"{name}/tokens"] = np.concatenate([np.array(s) for s in sents])
        for i, b in enumerate(batches):
            for k in ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens"):
                out[f"{name}/b{i}/{k}"] = npy(b[k])
            out[f"{name}/b{i}/num_sentence"] = np.array(b["num_sentence"])
    save("packing_text.npz", **out)


def asr_cases():
    rng = np.random.RandomState(7)
    F = 6
    out = {}
    for name, (B, T, drop, n) in {"mixed": (2, 24, False, 14), "droplast": (2, 20, True, 9)}.items():
        alens = [int(x) for x in rng.randint(2, 14, size=n)]
        tlens = [int(x) for x in rng.randint(1, 7, size=n)]
        alens[3] = 30  # too long for any row -> dropped (processing_touch_audio.py:169-170)
        feats = [rng.randn(a, F).astype(np.float32) for a in alens]
        ids = [[int(v) for v in rng.randint(3, 16, size=t)] for t in tlens]
        cfg = types.SimpleNamespace(dataset_batchsize=B, dataset_text_seqlen=T, dataset_audio_seqlen=T,
                                    audiofeat_num_mel_bins=F, audiofeat_stack_length=1,
                                    dataloader_drop_last_batch=drop)
        data = ({"audiofeat": torch.from_numpy(f), "input_ids": i} for f, i in zip(feats, ids))
        batches = list(ref_batch_asr(data, cfg, TOK))
        out[f"{name}/meta"] = np.array([B, T, int(drop), len(batches), F])
        out[f"{name}/alens"] = np.array(alens)
        out[f"{name}/tlens"] = np.array(tlens)
        out[f"{name}/feats"] = np.concatenate(feats, axis=0)
        out[f"{name}/tokens"] = np.concatenate([np.array(s) for s in ids])
        for i, b in enumerate(batches):
            for k in ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens",
                      "input_features", "shift_labels"):
                out[f"{name}/b{i}/{k}"] = npy(b[k])
            out[f"{name}/b{i}/num_sentence"] = np.array(b["num_sentence"])
    save("packing_asr.npz", **out)


def unpacked_asr_cases():
    """processing_touch_audio.py:217-428 run here: the two UNPACKED batchers (`batch_pairaudio_pairtext`, `batch_audio`) on
    seeded streams with a sample that is too long in the middle (it still moves the running maximum of the flush rule),
    drop_last on and off; `batch_audio` with a stand-in for the BEST-RQ tokenizer's `tokenize`."""
    from touchnet.models.touch_audio.processing_touch_audio import batch_audio as ref_batch_audio
    from touchnet.models.touch_audio.processing_touch_audio import batch_pairaudio_pairtext as ref_batch_pairs
    rng = np.random.RandomState(11)
    F = 5
    out = {}

    class Codes:
        def tokenize(self, feat):
            return [int(v) for v in (feat.sum(1) * 7.0).abs().long() % 50]
    for name, (B, T, drop, n) in {"mixed": (2, 24, False, 15), "droplast": (3, 20, True, 11), "tight": (1, 16, False, 9)}.items():
        alens = [int(x) for x in rng.randint(2, 13, size=n)]
        tlens = [int(x) for x in rng.randint(1, 6, size=n)]
        alens[4] = T + 9                      # longer than a row: skipped, but widens the running maximum
        feats = [rng.randn(a, F).astype(np.float32) for a in alens]
        ids = [[int(v) for v in rng.randint(3, 16, size=t)] for t in tlens]
        cfg = types.SimpleNamespace(dataset_batchsize=B, dataset_text_seqlen=T, dataset_audio_seqlen=T,
                                    dataloader_drop_last_batch=drop)
        out[f"{name}/meta"] = np.array([B, T, int(drop), F])
        out[f"{name}/alens"], out[f"{name}/tlens"] = np.array(alens), np.array(tlens)
        out[f"{name}/feats"] = np.concatenate(feats, axis=0)
        out[f"{name}/tokens"] = np.concatenate([np.array(s) for s in ids])
        for kind, fn, tok in (("pairs", ref_batch_pairs, TOK), ("audio", ref_batch_audio, Codes())):
            data = ({"audiofeat": torch.from_numpy(f), "input_ids": i} for f, i in zip(feats, ids))
            batches = list(fn(data, cfg, tok))
            out[f"{name}/{kind}/n"] = np.array(len(batches))
            for i, b in enumerate(batches):
                for k, v in b.items():
                    if k == "num_sentence":
                        out[f"{name}/{kind}/b{i}/{k}"] = np.array(v)
                    elif v is None:
                        out[f"{name}/{kind}/b{i}/{k}/none"] = np.array(1)
                    else:
                        out[f"{name}/{kind}/b{i}/{k}"] = npy(v)
    save("unpacked_asr.npz", **out)


# ------------------------------------------------------------------ loss
def ce_cases():
    out = {}
    # (1) the literal label vectors of tests/touchnet/utils/test_pack_loss.py:135-161, packed
    b1 = [-100, -100, 1, 2, 3]
    b2 = [4, -100, 3, 4, 6, -100, -100, 7]
    b3 = [-100, 6, 8]
    b4 = [-100, 7, 8, -100]
    b5 = [-100, -100, 7, 4, 2, 5]
    b6 = [5, 8, -100]
    order = [b1, b2, b3, b2, b6, b4, b5, b6]
    labels = torch.tensor(sum(order, []), dtype=torch.int64)[None]            # [1, 40]
    # per-sentence normaliser = number of valid labels (what the test divides by);
    # the packers would use len+1, the loss only sees a tensor
    sl = torch.tensor(sum([[sum(1 for v in s if v != -100)] * len(s) for s in order], []))[None]
    torch.manual_seed(0)
    logits = torch.randn(1, 40, 9, requires_grad=True)
    ps, pt = ref_ce(logits, labels, sl, 8)
    ps.backward()
    out.update({"pack/logits": npy(logits), "pack/labels": npy(labels), "pack/sentence_lens": npy(sl),
                "pack/num_sentence": np.array(8), "pack/loss_per_sample": npy(ps),
                "pack/loss_per_token": npy(pt), "pack/dlogits": npy(logits.grad)})
    # batch-split value of the same data (test_pack_loss.py:10-50) for the equivalence check
    # (2) mixed B=2,T=16,V=16 with packer-style sentence_lens and a bf16 copy
    torch.manual_seed(1)
    logits = (torch.randn(2, 16, 16) * 3).requires_grad_()
    labels = torch.randint(0, 16, (2, 16))
    labels[0, :3] = -100
    labels[1, 5:9] = -100
    labels[1, 15] = -100
    sl = torch.tensor([[4] * 4 + [7] * 7 + [5] * 5, [9] * 9 + [6] * 6 + [1]])
    ps, pt = ref_ce(logits, labels, sl, 11)
    ps.backward()
    out.update({"mixed/logits": npy(logits), "mixed/labels": npy(labels), "mixed/sentence_lens": npy(sl),
                "mixed/num_sentence": np.array(11), "mixed/loss_per_sample": npy(ps),
                "mixed/loss_per_token": npy(pt), "mixed/dlogits": npy(logits.grad)})
    # (3) nothing valid -> per-token 0 branch (cross_entropy.py:41-44)
    logits = torch.randn(1, 4, 5)
    labels = torch.full((1, 4), -100)
    ps, pt = ref_ce(logits, labels, torch.ones(1, 4, dtype=torch.int64), 1)
    out.update({"empty/logits": npy(logits), "empty/loss_per_sample": npy(ps), "empty/loss_per_token": npy(pt)})
    save("ce_loss.npz", **out)


# ------------------------------------------------------------------ doc mask
def docmask_cases():
    from transformers.integrations.flex_attention import make_flex_block_causal_mask
    docs = torch.tensor([[1, 1, 1, 2, 2, 2, 0], [1, 1, 2, 2, 2, 3, 3]])          # processing_llama.py:38-40
    rng = np.random.RandomState(3)
    big = []
    for _ in range(2):
        row, d = [], 1
        while len(row) < 150:
            row += [d] * int(rng.randint(1, 40))
            d += 1
        row = row[:150]
        row[140:] = [0] * 10
        big.append(row)
    out = {}
    for name, ids in {"small": docs, "big": torch.tensor(big)}.items():
        bm = make_flex_block_causal_mask(ids)
        B, T = ids.shape
        q = torch.arange(T)
        allow = torch.zeros(B, T, T, dtype=torch.bool)
        for b in range(B):
            for i in range(T):
                allow[b, i] = bm.mask_mod(torch.tensor(b), torch.tensor(0), torch.tensor(i), q)
        out[f"{name}/doc_ids"] = npy(ids)
        out[f"{name}/allow"] = npy(allow)
    save("docmask.npz", **out)


# ------------------------------------------------------------------ rope
def rope_cases():
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    out = {}
    for name, path in {"tiny": "tests/assets/config/tiny_llama.json",
                       "llama1b": "examples/text/pretrain/fineweb-edu/config/Llama-3_2-1B.json"}.items():
        cfg = LlamaConfig.from_json_file(f"{R.REF}/{path}")
        rot = LlamaRotaryEmbedding(cfg)
        # models/llama/__init__.py:23-27 re-derives inv_freq through `rope_init_fn`; transformers 5.x
        # (installed here) dropped that attribute and computes the same table in __init__.
        inv, scal = rot.inv_freq.clone(), rot.attention_scaling
        pos = torch.tensor([list(range(5)) + list(range(9)) + list(range(3)) + [0] * 3])
        cos, sin = rot(torch.zeros(1, 20, 4), pos)
        out[f"{name}/inv_freq"] = npy(inv)
        out[f"{name}/attention_scaling"] = np.array(scal)
        out[f"{name}/position_ids"] = npy(pos)
        out[f"{name}/cos"] = npy(cos)
        out[f"{name}/sin"] = npy(sin)
    save("rope.npz", **out)


# ------------------------------------------------------------------ tiny Llama / TouchAudio
def _allow4d(doc):
    T = doc.shape[1]
    q = torch.arange(T)
    allow = (q[:, None] >= q[None, :])[None] & (doc[:, :, None] > 0) & (doc[:, :, None] == doc[:, None, :])
    return torch.zeros(doc.shape[0], 1, T, T).masked_fill(~allow[:, None], torch.finfo(torch.float32).min)


def tiny_llama_case():
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig.from_json_file(f"{R.REF}/tests/assets/config/tiny_llama.json")
    cfg._attn_implementation = "eager"
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).float()
    # reference post_init (models/llama/__init__.py:19-36) = re-derive inv_freq + ones_ on norm weights;
    # both are what HF's own init already produced here (its `rope_init_fn` hook is gone in 5.x).
    rng = np.random.RandomState(11)
    sents = [[int(v) for v in rng.randint(3, 16, size=int(n))] for n in rng.randint(1, 13, size=14)]
    dcfg = types.SimpleNamespace(dataset_batchsize=4, dataset_text_seqlen=32, dataloader_drop_last_batch=False)
    batch = next(iter(ref_batch_text(({"input_ids": s} for s in sents), dcfg, TOK)))
    out = model(input_ids=batch["input_ids"], attention_mask=_allow4d(batch["attention_mask"]),
                position_ids=batch["position_ids"])
    ps, pt = ref_ce(out.logits, batch["labels"], batch["sentence_lens"], batch["num_sentence"])
    ps.backward()
    arrs = {f"param/{n}": npy(p) for n, p in model.named_parameters()}
    arrs.update({f"grad/{n}": npy(p.grad) for n, p in model.named_parameters()})
    for k in ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens"):
        arrs[f"batch/{k}"] = npy(batch[k])
    arrs["batch/num_sentence"] = np.array(batch["num_sentence"])
    arrs["logits"] = npy(out.logits)
    arrs["loss_per_sample"] = npy(ps)
    arrs["loss_per_token"] = npy(pt)
    save("tiny_llama.npz", **arrs)


def touch_audio_case():
    from transformers import LlamaConfig
    from touchnet.models.touch_audio.configuration_touch_audio import TouchAudioConfig
    from touchnet.models.touch_audio.modeling_touch_audio import TouchAudioForCausalLM
    tcfg = LlamaConfig.from_json_file(f"{R.REF}/tests/assets/config/tiny_llama.json").to_dict()
    F = 21
    cfg = TouchAudioConfig(text_config=tcfg, audio_config={"model_type": "touch_audio_projector", "input_size": F},
                           pad_token_id=0)
    cfg._attn_implementation = "eager"
    cfg.text_config._attn_implementation = "eager"
    torch.manual_seed(3)
    model = TouchAudioForCausalLM(cfg).float()
    rng = np.random.RandomState(5)
    n = 9
    feats = [rng.randn(int(a), F).astype(np.float32) for a in rng.randint(2, 12, size=n)]
    ids = [[int(v) for v in rng.randint(3, 16, size=int(t))] for t in rng.randint(1, 6, size=n)]
    dcfg = types.SimpleNamespace(dataset_batchsize=2, dataset_text_seqlen=48, dataset_audio_seqlen=48,
                                 audiofeat_num_mel_bins=F, audiofeat_stack_length=1,
                                 dataloader_drop_last_batch=False)
    data = ({"audiofeat": torch.from_numpy(f), "input_ids": i} for f, i in zip(feats, ids))
    batch = next(iter(ref_batch_asr(data, dcfg, TOK)))
    out = model(input_ids=batch["input_ids"], input_features=batch["input_features"],
                attention_mask=_allow4d(batch["attention_mask"]), position_ids=batch["position_ids"])
    ps, pt = ref_ce(out.logits, batch["labels"], batch["sentence_lens"], batch["num_sentence"])
    ps.backward()
    arrs = {f"param/{n}": npy(p) for n, p in model.named_parameters()}
    arrs.update({f"grad/{n}": npy(p.grad) for n, p in model.named_parameters()})
    for k in ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens", "input_features"):
        arrs[f"batch/{k}"] = npy(batch[k])
    arrs["batch/num_sentence"] = np.array(batch["num_sentence"])
    arrs["logits"] = npy(out.logits)
    arrs["loss_per_sample"] = npy(ps)
    arrs["loss_per_token"] = npy(pt)
    save("touch_audio.npz", **arrs)


# ------------------------------------------------------------------ Qwen2-Audio tower
def qwen2_audio_tower_case():
    from transformers.models.qwen2_audio.configuration_qwen2_audio import Qwen2AudioEncoderConfig
    from transformers.models.qwen2_audio.modeling_qwen2_audio import Qwen2AudioEncoder
    ref = R.load_file_as("ref_qwen2_audio_init", "touchnet/models/qwen2_audio/__init__.py")
    acfg = Qwen2AudioEncoderConfig(num_mel_bins=8, encoder_layers=2, encoder_attention_heads=4,
                                   encoder_ffn_dim=64, d_model=32, max_source_positions=10,
                                   dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    acfg._attn_implementation = "sdpa"
    torch.manual_seed(4)
    tower = Qwen2AudioEncoder(acfg).float().eval()
    with torch.no_grad():
        tower.embed_positions.weight.copy_(torch.randn_like(tower.embed_positions.weight) * 0.1)
    for layer in tower.layers:                       # qwen2_audio/__init__.py:191-192
        layer.self_attn.is_causal = True

    class AsTuple(torch.nn.Module):                  # 4.51.3 layers return a tuple, 5.x a tensor
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, h, mask=None, **kw):
            return (self.inner(h, mask),)
    tower.layers = torch.nn.ModuleList([AsTuple(l) for l in tower.layers])
    mel = torch.randn(2, 8, 52)                      # 52 -> 26 frames > 10 positions: exercises tiling
    out = ref.forward_audio_tower(tower, mel).last_hidden_state
    arrs = {f"param/{n.replace('.inner', '')}": npy(p) for n, p in tower.named_parameters()}
    arrs["embed_positions"] = npy(tower.embed_positions.weight)
    arrs["mel"] = npy(mel)
    arrs["out"] = npy(out)
    arrs["cfg"] = np.array([8, 2, 4, 64, 32, 10])
    save("qwen2_audio_tower.npz", **arrs)


# ------------------------------------------------------------------ frontend
def read_wav(path):
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2 and w.getnchannels() == 1
        sr = w.getframerate()
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    return pcm, sr


def frontend_cases():
    import librosa  # MagicMock; give it the slaney filter the way SURVEY.md §8c does
    from transformers.audio_utils import mel_filter_bank

    def fake_mel(sr, n_fft, n_mels):
        return mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels, min_frequency=0.0,
                               max_frequency=sr / 2.0, sampling_rate=sr, norm="slaney",
                               mel_scale="slaney").T.astype(np.float32)
    librosa.filters.mel = fake_mel
    out = {}
    # stack
    cases = {"arange": (np.arange(60, dtype=np.float32).reshape(20, 3), 7, 6)}
    rng = np.random.RandomState(9)
    for T in (1, 5, 6, 7, 13, 100):
        cases[f"r{T}_7_6"] = (rng.randn(T, 80).astype(np.float32), 7, 6)
    for T in (4, 9, 33, 257):
        cases[f"r{T}_5_4"] = (rng.randn(T, 80).astype(np.float32), 5, 4)
    cases["r50_4_4"] = (rng.randn(50, 16).astype(np.float32), 4, 4)
    cases["r31_1_1"] = (rng.randn(31, 16).astype(np.float32), 1, 1)
    for name, (x, stack, stride) in cases.items():
        cfg = types.SimpleNamespace(audiofeat_stack_length=stack, audiofeat_stride_length=stride,
                                    audiofeat_normalize=True)
        y = next(iter(ref_fn.audiofeat_stack(iter([{"audiofeat": torch.from_numpy(x)}]), cfg)))["audiofeat"]
        out[f"stack/{name}/x"] = x
        out[f"stack/{name}/y"] = npy(y)
        out[f"stack/{name}/ss"] = np.array([stack, stride])
    save("audiofeat_stack.npz", **out)

    out = {}
    wavdir = f"{R.REF}/tests/assets/dataset"
    for i, fn in enumerate(sorted(f for f in os.listdir(wavdir) if f.endswith(".wav"))):
        pcm, sr = read_wav(os.path.join(wavdir, fn))
        wav = torch.from_numpy(pcm.astype(np.float32) / 32768.0)[None]        # datapipe.py int16 -> /32768
        for n_mels in (80, 128):
            cfg = types.SimpleNamespace(audiofeat_padding=0, audiofeat_n_fft=400, audiofeat_hop_length=160,
                                        audiofeat_num_mel_bins=n_mels)
            y = next(iter(ref_fn.audio_compute_log_mel_spectrogram(
                iter([{"sample_rate": sr, "waveform": wav}]), cfg)))["audiofeat"]
            out[f"wav{i}/logmel{n_mels}"] = npy(y).astype(np.float32)
        out[f"wav{i}/pcm"] = pcm
        out[f"wav{i}/sr"] = np.array(sr)
    out["mel_filters_128"] = fake_mel(16000, 400, 128)
    save("logmel.npz", **out)


# ------------------------------------------------------------------ Kaldi fbank: independent third-party pin
def fbank_cases():
    """torchaudio (what functions.py:117-134 calls) is not in this image.  `transformers.audio_utils` ships an
    independent NumPy implementation written to match torchaudio.compliance.kaldi.fbank (the torchaudio-free fallback
    of its feature extractors: povey window, remove_dc_offset, 0.97 pre-emphasis, Kaldi mel scale triangularised in
    mel space, log of power with the fp32-epsilon floor).  Its output on the reference's two test wavs (int16-scale
    input, as the reference feeds torchaudio) is stored as the fixture the oracle's restatement is held to."""
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    out = {}
    wavdir = f"{R.REF}/tests/assets/dataset"
    for i, fn in enumerate(sorted(f for f in os.listdir(wavdir) if f.endswith(".wav"))):
        pcm, sr = read_wav(os.path.join(wavdir, fn))
        assert sr == 16000
        mel = mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20, max_frequency=8000,
                              sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
        fb = spectrogram(pcm.astype(np.float32), window_function(400, "povey", periodic=False), frame_length=400,
                         hop_length=160, fft_length=512, power=2.0, center=False, preemphasis=0.97, mel_filters=mel,
                         log_mel="log", mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
        out[f"wav{i}/fbank80"] = fb.astype(np.float32)
    save("kaldi_fbank_hf.npz", **out)


# ------------------------------------------------------------------ BEST-RQ tokenizer + audio-pretrain packer (§8f-4)
def bestrq_cases():
    from touchnet.models.touch_audio.processing_touch_audio import batch_audio_packed as ref_batch_audio
    from touchnet.tokenizer.tokenizer import BestRQTokenizer
    out = {}
    # (name, vocab, input_size, emb, seed): the recipe's shape (run.sh:107-109: vocab 1024, stack*mel) and the defaults
    for name, V, F, E, seed in (("recipe", 1024, 512, 16, 2026), ("default", 8192, 560, 16, 2026), ("small", 64, 24, 8, 7)):
        cfg = types.SimpleNamespace(tokenizer_bestrq_vocab_size=V, tokenizer_bestrq_input_size=F,
                                    tokenizer_bestrq_emb_size=E, tokenizer_bestrq_init_seed=seed,
                                    tokenizer_bestrq_init_method="default")
        tok = BestRQTokenizer(cfg)
        g = torch.Generator().manual_seed(100 + V)
        feat = torch.randn({1024: 300, 8192: 128, 64: 50}[V], F, generator=g)
        codes = tok.tokenize(feat)
        # the reference's own distances, for the near-tie margin of each frame (best vs second best)
        xs = torch.nn.functional.normalize(feat @ tok._quantizer, dim=-1, p=2, eps=1e-8)
        dist = torch.linalg.vector_norm(xs.unsqueeze(1) - tok._codebook.unsqueeze(0), dim=-1, ord=2)
        top2 = torch.topk(dist, 2, dim=-1, largest=False).values
        out[f"{name}/cfg"] = np.array([V, F, E, seed])
        out[f"{name}/feat"] = npy(feat)
        out[f"{name}/codes"] = np.asarray(codes, dtype=np.int64)
        out[f"{name}/margin"] = npy(top2[:, 1] - top2[:, 0])
        big = V > 1024                       # (the 8192-entry tables are stored subsampled: rows ::64 / ::16)
        out[f"{name}/quantizer"] = npy(tok._quantizer)[::16] if big else npy(tok._quantizer)
        out[f"{name}/codebook"] = npy(tok._codebook)[::64] if big else npy(tok._codebook)
        assert tok.vocab_size == V
    # packer: (B, T, drop_last, lengths incl. one over-long sample that must be skipped)
    cfg = types.SimpleNamespace(tokenizer_bestrq_vocab_size=64, tokenizer_bestrq_input_size=24,
                                tokenizer_bestrq_emb_size=8, tokenizer_bestrq_init_seed=7,
                                tokenizer_bestrq_init_method="default")
    tok = BestRQTokenizer(cfg)
    rng = np.random.RandomState(11)
    for name, B, T, drop, lens in (("overflow", 2, 40, False, [13, 20, 9, 41, 17, 30, 5, 26, 11]),
                                   ("droplast", 3, 32, True, [int(x) for x in rng.randint(4, 30, size=14)]),
                                   ("tail", 2, 64, False, [10, 12, 7])):
        dcfg = types.SimpleNamespace(dataset_batchsize=B, dataset_audio_seqlen=T, audiofeat_num_mel_bins=6,
                                     audiofeat_stack_length=4, dataloader_drop_last_batch=drop)
        g = torch.Generator().manual_seed(len(lens))
        feats = [torch.randn(n, 24, generator=g) for n in lens]
        batches = list(ref_batch_audio(({"audiofeat": f} for f in feats), dcfg, tok))
        out[f"pack/{name}/cfg"] = np.array([B, T, int(drop)])
        out[f"pack/{name}/lens"] = np.array(lens)
        out[f"pack/{name}/feats"] = np.concatenate([npy(f) for f in feats], 0)
        out[f"pack/{name}/n"] = np.array(len(batches))
        for i, bt in enumerate(batches):
            assert bt["input_ids"] is None and bt["shift_labels"] is bt["labels"]
            for k in ("input_features", "labels", "position_ids", "attention_mask", "sentence_lens"):
                out[f"pack/{name}/{i}/{k}"] = npy(bt[k])
            out[f"pack/{name}/{i}/num_sentence"] = np.array(bt["num_sentence"])
    save("bestrq.npz", **out)


# ------------------------------------------------------------------ TouchDataset reader + low-level datapipe (§8f-3)
def touchdataset_case():
    """Writes tests/golden/touchdataset/ with the reference's OWN writer classes from the reference's test assets
    (make_data.py's ffmpeg decode of a 16 kHz mono s16 wav is the identity on its samples, so the wave module
    stands in for it), checks the two md5s its test pins (tests/touchnet/bin/test_make_data.py:25-28), then records
    what the reference's TouchDataset / LowLevelTouchDatapipe read back under several configurations."""
    import hashlib
    import json
    import shutil
    import subprocess
    from touchnet.data.dataset import DType, IndexWriter, TouchDataset  # noqa: F401
    from touchnet.data.datapipe import LowLevelTouchDatapipe

    class Builder:                                    # = make_data.DataBuilder's calls into IndexWriter (:47-96)
        def __init__(self, bin_path, dtype):
            self.f, self.dtype, self.lens, self.docs = open(bin_path, "wb"), dtype, [], [0]

        def add(self, arr):
            a = np.array(arr, dtype=self.dtype)
            self.f.write(a.tobytes(order="C"))
            self.lens.append(a.size)
            self.docs.append(len(self.lens))

        def finalize(self, idx_path):
            self.f.close()
            with IndexWriter(idx_path, self.dtype) as w:
                w.write(self.lens, self.docs)

    root = os.path.join(HERE, "touchdataset")
    shutil.rmtree(root, ignore_errors=True)
    lines = [ln.strip() for ln in open(f"{R.REF}/tests/assets/dataset/data.jsonl")]

    def write_shards(save_dir, samples, per):
        shards = []
        for i in range(0, len(samples), per):
            d = "{}/{:09d}".format(save_dir, i // per)
            os.makedirs(d)
            a, m = Builder(f"{d}/audio.bin", np.int16), Builder(f"{d}/metainfo.bin", np.uint8)
            for meta, pcm in samples[i:i + per]:
                meta = dict(meta)
                meta["sample_rate"] = 16000
                a.add(pcm)
                m.add(np.frombuffer(json.dumps(meta, ensure_ascii=False).strip().encode("utf-8"), dtype=np.uint8))
            a.finalize(f"{d}/audio.idx")
            m.finalize(f"{d}/metainfo.idx")
            shards.append(d)
        return shards

    assets = []
    for ln in lines:
        meta = json.loads(ln)
        pcm, sr = read_wav(f"{R.REF}/{meta['wav']}")
        assert sr == 16000
        assets.append((meta, pcm))
    expect = {1: "05fe272d67459992748bbf5720c5a92e", 2: "93245372eca0dce2013c1e5bd393f17f"}
    for per in (1, 2):
        d = f"{root}/{per}sample_per_shard"
        write_shards(d, assets, per)
        cmd = (f"find {d} \\( -name '*.idx' -o -name '*.bin' \\) -type f -exec md5sum {{}} \\; | sort | "
               "cut -d ' ' -f1 | md5sum | awk '{print $1}'")
        got = subprocess.run(cmd, shell=True, capture_output=True, text=True).stdout.strip()
        assert got == expect[per], (per, got, expect[per])
        print(f"  {per}sample_per_shard md5 {got} == the reference test's constant")
    # a synthetic third dataset with segment annotations (the assets have none): 5 utterances in 2 shards
    rng = np.random.RandomState(3)
    synth = []
    for i in range(5):
        n = int(rng.randint(16000, 40000))
        pcm = (rng.randn(n) * 3000).astype(np.int16)
        segs, t = [], 0.0
        while t + 0.3 < n / 16000:
            e = min(t + float(rng.uniform(0.2, 0.9)), n / 16000)
            segs.append({"start": round(t, 2), "end": round(e, 2), "txt": f"utt{i}-seg{len(segs)}"})
            t = e
        synth.append(({"key": f"synth{i}", "wav": f"synth{i}.wav", "txt": f"text {i}", "info": {"segments": segs}}, pcm))
    write_shards(f"{root}/synthetic", synth, 3)

    out = {}

    def run(name, list_dirs, dp_rank=0, dp_world=1, state=None, limit=None, **over):
        lst = os.path.join(HERE, "_tmp_data.list")
        with open(lst, "w") as f:
            for d in list_dirs:
                f.write(f"{d} audio+metainfo\n")
        cfg = types.SimpleNamespace(datalist_path=lst, datalist_epoch=1, datalist_shuffling=False,
                                    datalist_sharding=False, dataset_mmap=True, dataset_shuffling=False,
                                    dataset_load_audio_via_segments=False, dataset_random_cut_audio=False,
                                    dataset_random_cut_audio_min_length_in_ms=5000,
                                    dataset_random_cut_audio_max_length_in_ms=3600000)
        for k, v in over.items():
            setattr(cfg, k, v)
        pipe = LowLevelTouchDatapipe(cfg, dp_rank, dp_world)
        if state:
            pipe.load_state_dict(state)
        keys, txts, lens, sums, heads, states = [], [], [], [], [], []
        for i, smp in enumerate(pipe):
            w = smp["waveform"]
            assert w.dtype == torch.float32 and w.dim() == 2 and w.shape[0] == 1
            pcm = (w[0].numpy() * 32768.0).astype(np.int64)
            keys.append(smp["key"]); txts.append(smp["txt"]); lens.append(pcm.size)
            sums.append(int(np.abs(pcm).sum())); heads.append(pcm[:4].tolist() + [0] * (4 - min(4, pcm.size)))
            states.append([pipe.epoch, pipe.consumed_lists, pipe.consumed_samples])
            if limit and i + 1 >= limit:
                break
        os.remove(lst)
        out[f"{name}/keys"] = np.array(keys); out[f"{name}/txts"] = np.array(txts)
        out[f"{name}/lens"] = np.array(lens); out[f"{name}/abs_sums"] = np.array(sums)
        out[f"{name}/heads"] = np.array(heads); out[f"{name}/states"] = np.array(states)
        print(f"  {name}: {len(keys)} samples {keys[:4]}")

    rel = lambda *p: os.path.join(root, *p)
    one = [rel("1sample_per_shard", "000000000"), rel("1sample_per_shard", "000000001")]
    two = [rel("2sample_per_shard", "000000000")]
    syn = [rel("synthetic", "000000000"), rel("synthetic", "000000001")]
    run("plain_1per", one)
    run("plain_2per", two)
    run("shuffled_2epochs", syn + one, datalist_epoch=2, datalist_shuffling=True, dataset_shuffling=True)
    run("sharded_rank1of2", syn + one, dp_rank=1, dp_world=2, datalist_sharding=True, datalist_shuffling=True)
    run("segments", syn, dataset_load_audio_via_segments=True, dataset_shuffling=True)
    run("random_cut", syn + two, dataset_random_cut_audio=True, dataset_random_cut_audio_min_length_in_ms=500,
        dataset_random_cut_audio_max_length_in_ms=1500)
    run("resumed", syn, state=dict(epoch=0, consumed_lists=0, consumed_samples=2), dataset_shuffling=True)
    # raw reader: partial reads
    ds = TouchDataset(two[0], True, "audio+metainfo")
    out["reader/len"] = np.array(len(ds))
    out["reader/idx0"] = np.array([int(v) for v in ds.get_idx(0, "audio")])
    out["reader/idx1"] = np.array([int(v) for v in ds.get_idx(1, "audio")])
    out["reader/partial"] = np.asarray(ds.get(1, "audio", offset=1000, length=64))
    out["reader/meta0"] = np.asarray(ds.get(0, "metainfo"))
    save("touchdataset.npz", **out)


# ------------------------------------------------------------------ boundary: TrainSpec schema + train.py call sites
def boundary_case():
    """Names and order of the reference's TrainSpec fields, the ParallelDims surface, apply_fsdp's parameters, and how
    touchnet/bin/train.py CALLS the hooks (positional count + keyword names per call site) — all read from the
    reference's syntax trees (importing these modules drags in torchdata / tensorboard) -> boundary.json (data only)."""
    import ast
    import json

    def tree(rel):
        return ast.parse(open(f"{R.REF}/{rel}").read())

    def klass(t, name):
        return next(n for n in ast.walk(t) if isinstance(n, ast.ClassDef) and n.name == name)

    spec = klass(tree("touchnet/utils/train_spec.py"), "TrainSpec")
    fields = [n.target.id for n in spec.body if isinstance(n, ast.AnnAssign)]
    required = [n.target.id for n in spec.body if isinstance(n, ast.AnnAssign) and n.value is None]
    calls = {}
    for node in ast.walk(tree("touchnet/bin/train.py")):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr in fields:
            v = node.func.value
            if isinstance(v, ast.Attribute) and v.attr == "train_spec":
                calls.setdefault(node.func.attr, []).append(
                    {"line": node.lineno, "n_positional": len(node.args),
                     "keywords": [k.arg for k in node.keywords if k.arg is not None],
                     "star_kwargs": any(k.arg is None for k in node.keywords)})
    dist_tree = tree("touchnet/utils/distributed.py")
    dims = klass(dist_tree, "ParallelDims")
    deco = lambda f: [d.id if isinstance(d, ast.Name) else getattr(d, "attr", "") for d in f.decorator_list]
    fsdp = next(n for n in ast.walk(tree("touchnet/models/helper_func.py"))
                if isinstance(n, ast.FunctionDef) and n.name == "apply_fsdp")
    mesh_names = sorted({c.value for n in ast.walk(dims) if isinstance(n, ast.Constant) and isinstance(n.value, str)
                         for c in [n] if c.value in ("pp", "dp_replicate", "dp_shard", "cp", "tp", "dp", "dp_shard_cp",
                                                     "dp_cp")})
    # the ORDER in which the trainer drives model / optimizer objects: attribute-call chains inside `train_step` and the
    # statements of the model set-up that follow `parallelize_fn` (what a data-parallel engine hidden behind the hooks
    # has to live with: a meta-device model at parallelize time, `.to(float32)` afterwards, no gradient visible to
    # clip_grad_norm_).  (name, line) pairs in source order.
    train = klass(tree("touchnet/bin/train.py"), "Trainer")

    def chain(n):
        parts = []
        while isinstance(n, ast.Attribute):
            parts.append(n.attr)
            n = n.value
        if isinstance(n, ast.Name):
            parts.append(n.id)
        return ".".join(reversed(parts))

    def calls_in(fn, wanted):
        found = []
        for node in ast.walk(fn):
            if isinstance(node, ast.Call):
                name = chain(node.func) if isinstance(node.func, ast.Attribute) else getattr(node.func, "id", "")
                if any(name.endswith(w) for w in wanted):
                    found.append([name, node.lineno])
        return sorted(found, key=lambda x: x[1])

    step_fn = next(n for n in train.body if isinstance(n, ast.FunctionDef) and n.name == "train_step")
    init_fn = next(n for n in train.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
    step_seq = calls_in(step_fn, ("optimizers.zero_grad", ".backward", "clip_grad_norm_", "optimizers.step",
                                  "lr_schedulers.step", "train_spec.loss_fn", "train_spec.acc_fn"))
    setup_seq = calls_in(init_fn, ("train_spec.parallelize_fn", "model.to_empty", "model.post_init",
                                   "train_spec.additional_post_init_fn", "model.train", "model.to",
                                   "train_spec.build_optimizers_fn", "train_spec.build_lr_schedulers_fn"))
    # the job config a reference run hands to the hooks: TrainConfig's field names with their literal defaults
    # (touchnet/bin/__init__.py:65-642).  What is NOT in this set cannot be set from the reference's command line
    # (HfArgumentParser rejects unknown flags) — e.g. there is no data-parallel engine switch.
    cfg_cls = klass(tree("touchnet/bin/__init__.py"), "TrainConfig")
    cfg_fields = {}
    for n in cfg_cls.body:
        if not isinstance(n, ast.AnnAssign) or n.target.id.startswith("_"):
            continue
        default = None
        if isinstance(n.value, ast.Call):
            for k in n.value.keywords:
                if k.arg == "default":
                    try:
                        default = ast.literal_eval(k.value)
                    except ValueError:
                        default = None
        cfg_fields[n.target.id] = default
    out = {"train_config_fields": cfg_fields,
           "train_spec_fields": fields, "train_spec_required": required, "train_py_calls": calls,
           "train_step_sequence": step_seq, "model_setup_sequence": setup_seq,
           "parallel_dims": {"fields": [n.target.id for n in dims.body if isinstance(n, ast.AnnAssign)],
                             "properties": sorted(f.name for f in dims.body if isinstance(f, ast.FunctionDef)
                                                  and ("property" in deco(f) or "cached_property" in deco(f))),
                             "mesh_dim_names": mesh_names},
           "apply_fsdp_params": [a.arg for a in fsdp.args.args]}
    path = os.path.join(HERE, "boundary.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote boundary.json: {os.path.getsize(path)} bytes")


# ------------------------------------------------------------------ Qwen2-Audio SFT samples (token-level content)
class _CharTokenizer:
    """Stand-in for the HF tokenizer the reference's processor carries (tokenizers are out of scope): special tokens
    <|...|> map to fixed ids, every other character to 10 + ord % 200.  Same call surface as used at
    processing_qwen2_audio.py:83-101."""
    SPECIAL = {"<|audio_bos|>": 3, "<|AUDIO|>": 4, "<|audio_eos|>": 5}
    eos_token_id, pad_token_id = 2, 0

    def convert_tokens_to_ids(self, tok):
        return self.SPECIAL[tok]

    def encode(self, text):
        ids, i = [], 0
        while i < len(text):
            for sp, v in self.SPECIAL.items():
                if text.startswith(sp, i):
                    ids.append(v)
                    i += len(sp)
                    break
            else:
                ids.append(10 + ord(text[i]) % 200)
                i += 1
        return ids

    def __call__(self, text, padding=False, return_tensors=None, add_special_tokens=True):
        ids = self.encode(text)
        if return_tensors == "pt":
            return {"input_ids": torch.tensor([ids])}
        return types.SimpleNamespace(input_ids=ids)


def qwen2_audio_data_case():
    from transformers import WhisperFeatureExtractor
    mod = R.load_file_as("ref_qwen2_audio_processing", "touchnet/models/qwen2_audio/processing_qwen2_audio.py")
    proc = types.SimpleNamespace(tokenizer=_CharTokenizer(), feature_extractor=WhisperFeatureExtractor(feature_size=128))
    rng = np.random.RandomState(11)
    durs = [0.31, 1.0, 2.503, 0.9999, 31.2, 4.0]                   # seconds; one clip longer than 30 s
    texts = ["ab", "hello world", "x", "the quick brown fox", "long audio", "tail"]
    samples = []
    for d, tx in zip(durs, texts):
        n = int(d * 16000)
        samples.append({"waveform": torch.from_numpy((rng.randn(1, n) * 0.05).astype(np.float32)), "txt": tx,
                        "sample_rate": 16000})
    samples[2]["instruct"] = "Translate:"
    cfg = types.SimpleNamespace(dataset_batchsize=1, dataset_text_seqlen=100000, dataloader_drop_last_batch=False,
                                audio_max_length_in_ms_for_filter=40000, text_min_length_in_tokens_for_filter=1,
                                text_max_length_in_tokens_for_filter=100000)
    out = {}
    for i, smp in enumerate(samples):                                # one reference batch per sample: no padding
        b = list(mod.dynamic_batch(iter([dict(smp)]), cfg, proc))
        assert len(b) == 1
        b = b[0]
        out[f"s{i}/n_samples"] = np.array(smp["waveform"].shape[1])
        for k in ("input_ids", "labels", "sentence_lens"):
            out[f"s{i}/{k}"] = npy(b[k][0])
        out[f"s{i}/feature_attention_mask_sum"] = np.array(int(b["feature_attention_mask"].sum()))
        feat = npy(b["input_features"][0])                           # [128, frames]
        out[f"s{i}/frames"] = np.array(feat.shape[1])
        out[f"s{i}/mel_head"] = feat[:, :40].astype(np.float32)      # first 40 frames + a strided sample of the rest
        out[f"s{i}/mel_strided"] = feat[:, ::97].astype(np.float32)
    out["wave_seed"] = np.array(11)
    out["durations"] = np.array(durs)
    out["texts"] = np.array(texts)
    # the stream through the reference's dynamic batching (:114-147): a budget that closes batches after 1-3 samples
    lens = [len(out[f"s{i}/input_ids"]) for i in range(len(samples))]
    for name, (bs, seqlen, drop) in {"dyn_a": (2, max(lens), False), "dyn_b": (1, 2 * sorted(lens)[2] + 1, True)}.items():
        cfg2 = types.SimpleNamespace(**{**vars(cfg), "dataset_batchsize": bs, "dataset_text_seqlen": seqlen,
                                        "dataloader_drop_last_batch": drop})
        batches = list(mod.dynamic_batch(iter([dict(s) for s in samples]), cfg2, proc))
        out[f"{name}/cfg"] = np.array([bs, seqlen, int(drop)])
        out[f"{name}/n"] = np.array(len(batches))
        for j, b in enumerate(batches):
            for k in ("input_ids", "attention_mask", "labels", "shift_labels", "sentence_lens", "feature_attention_mask"):
                out[f"{name}/b{j}/{k}"] = npy(b[k])
            out[f"{name}/b{j}/num_sentence"] = np.array(b["num_sentence"])
            feat = npy(b["input_features"])                          # [B, 128, frames]
            out[f"{name}/b{j}/feat_shape"] = np.array(feat.shape)
            out[f"{name}/b{j}/feat_strided"] = feat[:, ::8, ::97].astype(np.float32)
    save("qwen2_audio_data.npz", **out)

# ------------------------------------------------------------------ Kimi-Audio decoder (config E) — the reference's own module
def kimi_decoder_case():
    """MoonshotKimiaModel (modeling_kimi_audio.py:347-556: Qwen2 stack, mimo branch cloned after layer
    `kimia_mimo_transformer_from_layer_index`, two final norms) RUN from the reference's source with tiny widths, plus the
    embedding sum and the two heads of MoonshotKimiaForCausalLM.forward (:1026-1068, three lines reproduced here because
    that class also builds the Whisper encoder and the frozen VQ tokenizer, which are out of scope).  Eager attention with
    the 4-D document-causal additive mask (`_update_causal_mask` passes a 4-D mask through untouched, :690-692).
    Import recipe and the two API-drift adapters: _ref_import.py::load_kimi_modeling / adapt_decoder_layers_to_4_51."""
    mk = R.load_kimi_modeling()
    from touchnet.models.kimi_audio.configuration_kimi_audio import KimiAudioConfig
    kw = dict(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=4, num_attention_heads=4,
              num_key_value_heads=2, rms_norm_eps=1e-6, rope_theta=1e6, kimia_mimo_layers=2,
              kimia_mimo_transformer_from_layer_index=1, use_whisper_feature=False, use_cache=False, pad_token_id=None,
              initializer_range=0.1)
    cfg = KimiAudioConfig(**kw)
    cfg._attn_implementation = "eager"
    # (5.x reads a per-layer attention type from the config; the mimo layers carry indices L .. L + L_mimo - 1: full
    # attention for all of them = what 4.51.3's Qwen2Attention does without `use_sliding_window`)
    cfg.layer_types = ["full_attention"] * (kw["num_hidden_layers"] + kw["kimia_mimo_layers"])
    torch.manual_seed(0)
    model = mk.MoonshotKimiaModel(cfg).float().eval()
    R.adapt_decoder_layers_to_4_51(list(model.layers) + list(model.mimo_layers))
    lm_head = torch.nn.Linear(64, 64, bias=False)                       # (:860-861: two bias-free heads)
    mimo_output = torch.nn.Linear(64, 64, bias=False)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in model.named_parameters():                          # give the q/k/v biases non-zero values
            if n.endswith("bias"):
                p.normal_(std=0.05, generator=g)
    B, T = 2, 48
    a, t = torch.randint(0, 64, (B, T), generator=g), torch.randint(0, 64, (B, T), generator=g)
    doc = torch.cat([torch.ones(B, 20), 2 * torch.ones(B, 20), torch.zeros(B, 8)], 1).long()
    pos = torch.cat([torch.arange(20), torch.arange(20), torch.zeros(8, dtype=torch.long)]).repeat(B, 1)
    emb = model.get_input_embeddings()
    inputs_embeds = emb(a) + emb(t)                                     # :1030-1033
    out = model(input_ids=None, inputs_embeds=inputs_embeds, attention_mask=_allow4d(doc), position_ids=pos,
                use_cache=False, return_dict=True)
    hidden, mimo = out.last_hidden_state                                # :549-550
    text_logits, audio_logits = lm_head(hidden), mimo_output(mimo)      # :1060-1061
    labels = torch.where(doc > 0, torch.randint(0, 64, (B, T), generator=g), torch.full((B, T), -100))
    sl = torch.where(doc > 0, torch.full((B, T), 20), torch.ones(B, T, dtype=torch.long))
    ps, pt = ref_ce(text_logits, labels, sl, 4)
    ps.backward()
    arrs = {f"param/model.{n}": npy(p) for n, p in model.named_parameters()}
    arrs["param/lm_head.weight"], arrs["param/mimo_output.weight"] = npy(lm_head.weight), npy(mimo_output.weight)
    arrs.update({f"grad/model.{n}": npy(p.grad) for n, p in model.named_parameters() if p.grad is not None})
    arrs["grad/lm_head.weight"] = npy(lm_head.weight.grad)
    arrs.update({"batch/audio_input_ids": npy(a), "batch/text_input_ids": npy(t), "batch/attention_mask": npy(doc),
                 "batch/position_ids": npy(pos), "batch/labels": npy(labels), "batch/sentence_lens": npy(sl),
                 "text_logits": npy(text_logits), "audio_logits": npy(audio_logits), "loss_per_sample": npy(ps),
                 "loss_per_token": npy(pt)})
    arrs["config_json"] = np.array(str({k: v for k, v in kw.items() if k not in ("use_whisper_feature", "use_cache", "pad_token_id")}))
    save("kimi_decoder.npz", **arrs)


# ------------------------------------------------------------------ device-shaped model fixtures
# The fixtures above use head_dim 8 / 16: shapes the MI355X attention kernels do not take.  These are the SAME reference
# modules at the smallest widths the kernels accept (head_dim 64), run in float32 on bf16-ROUNDED weights (what the device
# model holds), so that a `-m gpu` test can load the fixture into the product model on the MI355X and compare with the
# reference's own outputs without the product's wiring in between.  Weights are stored as bf16 bits (uint16), gradients
# as float16 (their tolerance on the device is two orders of magnitude wider).
def _bf16_round_(model):
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.bfloat16().float())


def _bits(t):
    return t.detach().bfloat16().view(torch.int16).cpu().numpy().view(np.uint16)


def _dev_text_config():
    from transformers import LlamaConfig
    cfg = LlamaConfig.from_json_file(f"{R.REF}/tests/assets/config/tiny_llama.json")
    cfg.update(dict(vocab_size=64, hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                    num_key_value_heads=1, head_dim=64))
    return cfg


def _model_arrays(model, out, batch, ps, pt, keys):
    arrs = {f"param/{n}": _bits(p) for n, p in model.named_parameters()}
    arrs.update({f"grad/{n}": npy(p.grad).astype(np.float16) for n, p in model.named_parameters() if p.grad is not None})
    for k in keys:
        arrs[f"batch/{k}"] = npy(batch[k])
    arrs["batch/num_sentence"] = np.array(batch["num_sentence"])
    arrs["logits"] = npy(out.logits)
    arrs["loss_per_sample"], arrs["loss_per_token"] = npy(ps), npy(pt)
    return arrs


def tiny_llama_dev_case():
    from transformers import LlamaForCausalLM
    cfg = _dev_text_config()
    cfg._attn_implementation = "eager"
    torch.manual_seed(20)
    model = LlamaForCausalLM(cfg).float()
    _bf16_round_(model)
    rng = np.random.RandomState(21)
    sents = [[int(v) for v in rng.randint(3, 64, size=int(n))] for n in rng.randint(4, 60, size=12)]
    dcfg = types.SimpleNamespace(dataset_batchsize=2, dataset_text_seqlen=128, dataloader_drop_last_batch=False)
    batch = next(iter(ref_batch_text(({"input_ids": s} for s in sents), dcfg, TOK)))
    out = model(input_ids=batch["input_ids"], attention_mask=_allow4d(batch["attention_mask"]),
                position_ids=batch["position_ids"])
    ps, pt = ref_ce(out.logits, batch["labels"], batch["sentence_lens"], batch["num_sentence"])
    ps.backward()
    arrs = _model_arrays(model, out, batch, ps, pt, ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens"))
    # (the json's 4.51-style rope_theta / rope_scaling: transformers 5.x folds both into `rope_parameters`)
    rp = dict(cfg.rope_parameters)
    theta = rp.pop("rope_theta")
    desc = {k: getattr(cfg, k) for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers",
                                         "num_attention_heads", "num_key_value_heads", "head_dim", "rms_norm_eps",
                                         "tie_word_embeddings")}
    desc.update(rope_theta=theta, rope_scaling=rp)
    arrs["config_json"] = np.array(str(desc))
    save("tiny_llama_dev.npz", **arrs)


def touch_audio_dev_case():
    from touchnet.models.touch_audio.configuration_touch_audio import TouchAudioConfig
    from touchnet.models.touch_audio.modeling_touch_audio import TouchAudioForCausalLM
    F = 16
    cfg = TouchAudioConfig(text_config=_dev_text_config().to_dict(),
                           audio_config={"model_type": "touch_audio_projector", "input_size": F}, pad_token_id=0)
    cfg._attn_implementation = "eager"
    cfg.text_config._attn_implementation = "eager"
    torch.manual_seed(22)
    model = TouchAudioForCausalLM(cfg).float()
    _bf16_round_(model)
    rng = np.random.RandomState(23)
    n = 10
    feats = [torch.from_numpy(rng.randn(int(a), F).astype(np.float32)).bfloat16().float().numpy()
             for a in rng.randint(6, 40, size=n)]
    ids = [[int(v) for v in rng.randint(3, 64, size=int(t))] for t in rng.randint(2, 12, size=n)]
    dcfg = types.SimpleNamespace(dataset_batchsize=2, dataset_text_seqlen=128, dataset_audio_seqlen=128,
                                 audiofeat_num_mel_bins=F, audiofeat_stack_length=1, dataloader_drop_last_batch=False)
    data = ({"audiofeat": torch.from_numpy(f), "input_ids": i} for f, i in zip(feats, ids))
    batch = next(iter(ref_batch_asr(data, dcfg, TOK)))
    out = model(input_ids=batch["input_ids"], input_features=batch["input_features"],
                attention_mask=_allow4d(batch["attention_mask"]), position_ids=batch["position_ids"])
    ps, pt = ref_ce(out.logits, batch["labels"], batch["sentence_lens"], batch["num_sentence"])
    ps.backward()
    arrs = _model_arrays(model, out, batch, ps, pt,
                         ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens", "input_features"))
    arrs["input_size"] = np.array(F)
    save("touch_audio_dev.npz", **arrs)


def qwen2_audio_tower_dev_case():
    from transformers.models.qwen2_audio.configuration_qwen2_audio import Qwen2AudioEncoderConfig
    from transformers.models.qwen2_audio.modeling_qwen2_audio import Qwen2AudioEncoder
    ref = R.load_file_as("ref_qwen2_audio_init", "touchnet/models/qwen2_audio/__init__.py")
    dims = dict(num_mel_bins=16, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256, d_model=128,
                max_source_positions=50)
    acfg = Qwen2AudioEncoderConfig(**dims, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    acfg._attn_implementation = "sdpa"
    torch.manual_seed(24)
    tower = Qwen2AudioEncoder(acfg).float().eval()
    with torch.no_grad():
        tower.embed_positions.weight.copy_(torch.randn_like(tower.embed_positions.weight) * 0.1)
    _bf16_round_(tower)
    for layer in tower.layers:                       # qwen2_audio/__init__.py:191-192
        layer.self_attn.is_causal = True

    class AsTuple(torch.nn.Module):                  # 4.51.3 layers return a tuple, 5.x a tensor
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, h, mask=None, **kw):
            return (self.inner(h, mask),)
    tower.layers = torch.nn.ModuleList([AsTuple(l) for l in tower.layers])
    mel = torch.randn(3, 16, 100).bfloat16().float()  # 100 mel frames -> 50 positions -> 25 tokens per clip
    out = ref.forward_audio_tower(tower, mel).last_hidden_state
    arrs = {f"param/{n.replace('.inner', '')}": _bits(p) for n, p in tower.named_parameters()}
    arrs["embed_positions"] = _bits(tower.embed_positions.weight)
    arrs["mel"], arrs["out"] = npy(mel), npy(out)
    arrs["config_json"] = np.array(str(dims))
    save("qwen2_audio_tower_dev.npz", **arrs)


def kimi_decoder_dev_case():
    mk = R.load_kimi_modeling()
    from touchnet.models.kimi_audio.configuration_kimi_audio import KimiAudioConfig
    kw = dict(vocab_size=64, hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
              num_key_value_heads=1, rms_norm_eps=1e-6, rope_theta=1e6, kimia_mimo_layers=1,
              kimia_mimo_transformer_from_layer_index=0, use_whisper_feature=False, use_cache=False, pad_token_id=None,
              initializer_range=0.1)
    cfg = KimiAudioConfig(**kw)
    cfg._attn_implementation = "eager"
    cfg.layer_types = ["full_attention"] * (kw["num_hidden_layers"] + kw["kimia_mimo_layers"])
    torch.manual_seed(25)
    model = mk.MoonshotKimiaModel(cfg).float().eval()
    R.adapt_decoder_layers_to_4_51(list(model.layers) + list(model.mimo_layers))
    lm_head = torch.nn.Linear(128, 64, bias=False)
    mimo_output = torch.nn.Linear(128, 64, bias=False)
    g = torch.Generator().manual_seed(26)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.normal_(std=0.05, generator=g)
    for m in (model, lm_head, mimo_output):
        _bf16_round_(m)
    B, T = 2, 128
    a, t = torch.randint(0, 64, (B, T), generator=g), torch.randint(0, 64, (B, T), generator=g)
    doc = torch.cat([torch.ones(B, 50), 2 * torch.ones(B, 60), torch.zeros(B, 18)], 1).long()
    pos = torch.cat([torch.arange(50), torch.arange(60), torch.zeros(18, dtype=torch.long)]).repeat(B, 1)
    emb = model.get_input_embeddings()
    inputs_embeds = emb(a) + emb(t)
    out = model(input_ids=None, inputs_embeds=inputs_embeds, attention_mask=_allow4d(doc), position_ids=pos,
                use_cache=False, return_dict=True)
    hidden, mimo = out.last_hidden_state
    text_logits, audio_logits = lm_head(hidden), mimo_output(mimo)
    labels = torch.where(doc > 0, torch.randint(0, 64, (B, T), generator=g), torch.full((B, T), -100))
    sl = torch.where(doc == 1, torch.full((B, T), 50), torch.where(doc == 2, torch.full((B, T), 60), torch.ones(B, T, dtype=torch.long)))
    ps, pt = ref_ce(text_logits, labels, sl, 4)
    ps.backward()
    arrs = {f"param/model.{n}": _bits(p) for n, p in model.named_parameters()}
    arrs["param/lm_head.weight"], arrs["param/mimo_output.weight"] = _bits(lm_head.weight), _bits(mimo_output.weight)
    arrs.update({f"grad/model.{n}": npy(p.grad).astype(np.float16) for n, p in model.named_parameters() if p.grad is not None})
    arrs["grad/lm_head.weight"] = npy(lm_head.weight.grad).astype(np.float16)
    arrs.update({"batch/audio_input_ids": npy(a), "batch/text_input_ids": npy(t), "batch/attention_mask": npy(doc),
                 "batch/position_ids": npy(pos), "batch/labels": npy(labels), "batch/sentence_lens": npy(sl),
                 "text_logits": npy(text_logits), "audio_logits": npy(audio_logits), "loss_per_sample": npy(ps),
                 "loss_per_token": npy(pt)})
    arrs["config_json"] = np.array(str({k: v for k, v in kw.items() if k not in ("use_whisper_feature", "use_cache", "pad_token_id")}))
    save("kimi_decoder_dev.npz", **arrs)


def kimi_audio_input_case():
    """MoonshotKimiaForCausalLM.prepare_audio_input_embs (modeling_kimi_audio.py:933-985) RUN from the reference's source:
    its CustomWhisperEncoder (transformers' WhisperEncoder) + VQAdaptor + embedding sum x sqrt(2) + masked_scatter between
    the media markers, at kernel-sized tiny widths (head_dim 64) on bf16-rounded weights.  The method is called unbound on
    a stand-in `self` that carries exactly the attributes it reads; the frozen GLM-4-voice speech tokenizer
    (`self.speech_tokenizer`, out of scope) is a stub returning fixed ids.  A linear read-out of the result gives
    gradients for every parameter on the path (encoder, adaptor, embedding)."""
    mk = R.load_kimi_modeling()
    from transformers import WhisperConfig
    H, d, S4 = 128, 128, 4
    dims = dict(num_mel_bins=16, d_model=d, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256,
                max_source_positions=40)
    wcfg = WhisperConfig(**dims, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    wcfg._attn_implementation = "sdpa"
    torch.manual_seed(30)
    enc = mk.CustomWhisperEncoder(wcfg).float().eval()
    with torch.no_grad():                                 # (CustomWhisperEncoder skips HF's init: draw everything here)
        for n, p in enc.named_parameters():
            if p.dim() > 1:
                p.normal_(std=0.05)
            elif n.endswith("bias"):
                p.normal_(std=0.02)
            else:
                p.fill_(1.0)
        enc.embed_positions.weight.copy_(torch.randn_like(enc.embed_positions.weight) * 0.1)
    kcfg = types.SimpleNamespace(kimia_adaptor_input_dim=S4 * d, hidden_size=H, rms_norm_eps=1e-6, kimia_token_offset=96)
    adaptor = mk.VQAdaptor(kcfg).float()
    emb = torch.nn.Embedding(160, H)
    with torch.no_grad():
        emb.weight.normal_(std=0.1)
        for p in adaptor.parameters():
            if p.dim() > 1:
                p.normal_(std=0.05)
    for m in (enc, adaptor, emb):
        _bf16_round_(m)
    g = torch.Generator().manual_seed(31)
    B, T, begin, end = 2, 32, 90, 91
    a = torch.randint(0, 80, (B, T), generator=g)
    spans = [(3, 10), (5, 7)]                              # (position of the begin marker, frames between the markers)
    for b, (p0, n) in enumerate(spans):
        a[b, p0], a[b, p0 + n + 1] = begin, end
    feats = torch.randn(B, 16, 80, generator=g).bfloat16().float()
    ids = torch.randint(0, 60, (B, 10), generator=g)

    class Tok:                                             # stand-in for the frozen WhisperVQEncoder (`:957-963`)
        def __call__(self, input_features=None, attention_mask=None, return_dict=True):
            return ids.clone()
    fake = types.SimpleNamespace(speech_encoder=enc, speech_tokenizer=Tok(), config=kcfg,
                                 model=types.SimpleNamespace(vq_adaptor=adaptor, kimia_media_begin=begin, kimia_media_end=end),
                                 get_input_embeddings=lambda: emb)
    fake.create_mask_between_markers = lambda **kw: mk.MoonshotKimiaForCausalLM.create_mask_between_markers(fake, **kw)
    out = mk.MoonshotKimiaForCausalLM.prepare_audio_input_embs(fake, audio_input_ids=a, audio_input_embs=emb(a),
                                                               whisper_input_features=feats, whisper_attention_mask=None)
    readout = torch.randn(B, T, H, generator=g) * 0.1
    (out * readout).sum().backward()
    arrs = {f"param/speech_encoder.{n}": _bits(p) for n, p in enc.named_parameters()}
    arrs["param/speech_encoder.embed_positions.weight"] = _bits(enc.embed_positions.weight)
    arrs.update({f"param/model.vq_adaptor.{n}": _bits(p) for n, p in adaptor.named_parameters()})
    arrs["param/model.embed_tokens.weight"] = _bits(emb.weight)
    arrs.update({f"grad/speech_encoder.{n}": npy(p.grad).astype(np.float16) for n, p in enc.named_parameters() if p.grad is not None})
    arrs.update({f"grad/model.vq_adaptor.{n}": npy(p.grad).astype(np.float16) for n, p in adaptor.named_parameters()})
    arrs["grad/model.embed_tokens.weight"] = npy(emb.weight.grad).astype(np.float16)
    arrs.update({"audio_input_ids": npy(a), "whisper_input_features": npy(feats), "speech_tokenizer_ids": npy(ids),
                 "readout": npy(readout), "out": npy(out), "markers": np.array([begin, end]),
                 "config_json": np.array(str(dict(dims, hidden_size=H, kimia_adaptor_input_dim=S4 * d, rms_norm_eps=1e-6,
                                                  kimia_token_offset=96, vocab_size=160)))})
    save("kimi_audio_input.npz", **arrs)


class _KimiTokenizer:
    """Stand-in for the reference's BaseTokenizer over the Kimi vocabulary (tokenizers are out of scope): the special
    tokens <|...|> of the two prompt templates map to fixed ids, every other character to 10 + ord % 200; same call
    surface as used at processing_kimi_audio.py:82-110 (`tokenize(text, add_special_tokens=False)`, `.pad`)."""
    SPECIAL = {"<|im_kimia_user_msg_start|>": 300, "<|im_kimia_text_blank|>": 301, "<|im_media_begin|>": 302,
               "<|im_media_end|>": 303, "<|im_kimia_speech_ct_id|>": 304, "<|im_msg_end|>": 305,
               "<|im_kimia_assistant_msg_start|>": 306, "<|im_kimia_text_eos|>": 307}
    pad = 0

    def tokenize(self, text, add_special_tokens=False):
        ids, i = [], 0
        while i < len(text):
            for sp, v in self.SPECIAL.items():
                if text.startswith(sp, i):
                    ids.append(v)
                    i += len(sp)
                    break
            else:
                ids.append(10 + ord(text[i]) % 200)
                i += 1
        return ids


def kimi_audio_data_case():
    """The reference's Kimi-Audio batcher (processing_kimi_audio.py:37-224) RUN on synthetic waveforms with HF's
    WhisperFeatureExtractor and the stand-in tokenizer above: token streams, labels, sentence lengths, masks, the
    batching rule (two batches + the last one) and a sample of the log-mel features."""
    from transformers import WhisperFeatureExtractor
    sys.modules.setdefault("touchnet.data", types.ModuleType("touchnet.data")).DataConfig = object
    dp = types.ModuleType("touchnet.data.datapipe")
    dp.LowLevelTouchDatapipe = dp.MidLevelTouchDatapipe = object
    sys.modules.setdefault("touchnet.data.datapipe", dp)
    tk = types.ModuleType("touchnet.tokenizer.tokenizer")
    tk.BaseTokenizer = object
    sys.modules.setdefault("touchnet.tokenizer", types.ModuleType("touchnet.tokenizer"))
    sys.modules.setdefault("touchnet.tokenizer.tokenizer", tk)
    mod = R.load_file_as("ref_kimi_processing", "touchnet/models/kimi_audio/processing_kimi_audio.py")
    proc = WhisperFeatureExtractor(feature_size=128)
    rng = np.random.RandomState(13)
    durs = [0.31, 1.0, 2.503, 0.9999, 4.0, 29.99, 0.05]
    texts = ["ab", "hello world", "x", "the quick brown fox", "tail", "long", "tiny"]
    samples = []
    for d, tx in zip(durs, texts):
        n = int(d * 16000)
        samples.append({"waveform": torch.from_numpy((rng.randn(1, n) * 0.05).astype(np.float32)), "txt": tx})
    samples[2]["instruct"] = "Translate:"
    cfg = types.SimpleNamespace(dataset_batchsize=2, dataset_text_seqlen=100, dataloader_drop_last_batch=False,
                                text_min_length_in_tokens_for_filter=1, text_max_length_in_tokens_for_filter=420)
    batches = list(mod.dynamic_batch(iter([dict(s) for s in samples]), cfg, proc, _KimiTokenizer()))
    out = {"n_batches": np.array(len(batches)), "durations": np.array(durs), "texts": np.array(texts), "wave_seed": np.array(13)}
    for i, b in enumerate(batches):
        for k in ("text_input_ids", "audio_input_ids", "attention_mask", "labels", "sentence_lens"):
            out[f"b{i}/{k}"] = npy(b[k])
        out[f"b{i}/num_sentence"] = np.array(b["num_sentence"])
        out[f"b{i}/whisper_attention_mask_sum"] = npy(b["whisper_attention_mask"].sum(1))
        feat = npy(b["whisper_input_features"])                      # [n, 128, 3000]
        out[f"b{i}/mel_head"] = feat[:, :, :40].astype(np.float32)
        out[f"b{i}/mel_strided"] = feat[:, :, ::97].astype(np.float32)
    save("kimi_audio_data.npz", **out)


def audiofeat_augment_case():
    """touchnet/data/functions.py:193-255 run here: every stage alone and the reference's chain spec_aug -> spec_sub ->
    spec_trim (processing_touch_audio.py:468-473; chained generators, two utterances per chain so that the draws of the
    second utterance follow the first one's through all stages), under `random.seed(seed)`.  Stored: inputs, the option
    values, the seed and the outputs — the twins replay the same global `random` stream."""
    import random
    rng = np.random.RandomState(17)
    out, names = {}, []

    def cfg_of(**kw):
        base = dict(audiofeat_spec_aug=False, audiofeat_spec_aug_num_t_mask=2, audiofeat_spec_aug_num_f_mask=2,
                    audiofeat_spec_aug_max_t=50, audiofeat_spec_aug_max_f=10, audiofeat_spec_sub=False,
                    audiofeat_spec_sub_num_t_sub=3, audiofeat_spec_sub_max_t=30, audiofeat_spec_trim=False,
                    audiofeat_spec_trim_max_t=20)
        base.update(kw)
        return types.SimpleNamespace(**base)

    def run(name, shapes, seed, **kw):
        cfg = cfg_of(**kw)
        xs = [rng.randn(T, F).astype(np.float32) for T, F in shapes]
        data = iter([{"audiofeat": torch.from_numpy(x.copy())} for x in xs])
        if cfg.audiofeat_spec_aug:
            data = ref_fn.audiofeat_spec_aug(data, cfg)
        if cfg.audiofeat_spec_sub:
            data = ref_fn.audiofeat_spec_sub(data, cfg)
        if cfg.audiofeat_spec_trim:
            data = ref_fn.audiofeat_spec_trim(data, cfg)
        random.seed(seed)
        ys = [npy(smp["audiofeat"]) for smp in data]
        names.append(name)
        out[f"{name}/seed"] = np.array(seed)
        out[f"{name}/cfg"] = np.array([int(cfg.audiofeat_spec_aug), cfg.audiofeat_spec_aug_num_t_mask,
                                       cfg.audiofeat_spec_aug_num_f_mask, cfg.audiofeat_spec_aug_max_t,
                                       cfg.audiofeat_spec_aug_max_f, int(cfg.audiofeat_spec_sub),
                                       cfg.audiofeat_spec_sub_num_t_sub, cfg.audiofeat_spec_sub_max_t,
                                       int(cfg.audiofeat_spec_trim), cfg.audiofeat_spec_trim_max_t])
        out[f"{name}/n"] = np.array(len(xs))
        for i, (x, y) in enumerate(zip(xs, ys)):
            out[f"{name}/x{i}"], out[f"{name}/y{i}"] = x, y

    # the wenetspeech ASR recipe's values (examples/audio/sft/asr/wenetspeech/run.sh:261-270)
    run("recipe_aug_sub", [(217, 80), (123, 80)], 2025, audiofeat_spec_aug=True, audiofeat_spec_sub=True)
    run("all_three", [(300, 24), (64, 80), (1000, 8)], 7, audiofeat_spec_aug=True, audiofeat_spec_sub=True,
        audiofeat_spec_trim=True)
    run("aug_only", [(200, 32), (37, 23)], 1, audiofeat_spec_aug=True)
    run("sub_only", [(200, 16), (90, 16)], 2, audiofeat_spec_sub=True)
    run("trim_only", [(200, 8), (30, 8), (41, 8)], 3, audiofeat_spec_trim=True)     # 30 frames: never trimmed by >= 15
    # stripes wider than the matrix, many of them (overlaps, clipping at the ends), one-frame utterances
    run("wide", [(20, 8), (1, 8), (2, 8)], 4, audiofeat_spec_aug=True, audiofeat_spec_aug_num_t_mask=5,
        audiofeat_spec_aug_num_f_mask=4, audiofeat_spec_aug_max_t=40, audiofeat_spec_aug_max_f=12, audiofeat_spec_sub=True,
        audiofeat_spec_sub_num_t_sub=8, audiofeat_spec_sub_max_t=15, audiofeat_spec_trim=True, audiofeat_spec_trim_max_t=3)
    for seed in range(5, 9):
        run(f"seed{seed}", [(150, 12), (151, 12)], seed, audiofeat_spec_aug=True, audiofeat_spec_sub=True,
            audiofeat_spec_trim=True, audiofeat_spec_trim_max_t=100)
    out["names"] = np.array(names)
    save("audiofeat_augment.npz", **out)


def qwen2_audio_model_dev_case():
    """The reference's Qwen2-Audio forward (touchnet/models/qwen2_audio/__init__.py:134-249: feature lengths from
    `feature_attention_mask`, tower, compaction of the valid audio rows, `masked_scatter`, causal language model) RUN on a batch
    in the reference's own (unpacked, right-padded) format, + the reference loss and every gradient.  transformers 5.x moved
    Qwen2AudioForConditionalGeneration's submodules (`.model.*`, `.lm_head`); the reference's forward is written against
    4.51.3's layout (`.audio_tower`, `.multi_modal_projector`, `.language_model` = Qwen2ForCausalLM), so the three HF modules
    are assembled in that layout in a plain nn.Module and handed to the reference's function as `self` — every statement of
    the reference's forward and forward_audio_tower executes.  head_dim 64, bf16-representable weights (the `-m gpu` twin)."""
    from transformers import Qwen2Config, Qwen2ForCausalLM
    from transformers.models.qwen2_audio.configuration_qwen2_audio import Qwen2AudioConfig, Qwen2AudioEncoderConfig
    from transformers.models.qwen2_audio.modeling_qwen2_audio import Qwen2AudioEncoder, Qwen2AudioMultiModalProjector
    ref = R.load_file_as("ref_qwen2_audio_init", "touchnet/models/qwen2_audio/__init__.py")
    adims = dict(num_mel_bins=16, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=128, d_model=128,
                 max_source_positions=50)
    tdims = dict(vocab_size=128, hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                 num_key_value_heads=1, head_dim=64, rope_theta=1e6, tie_word_embeddings=False, max_position_embeddings=512,
                 rms_norm_eps=1e-6)
    acfg = Qwen2AudioEncoderConfig(**adims, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    tcfg = Qwen2Config(**tdims)
    cfg = Qwen2AudioConfig(audio_config=acfg, text_config=tcfg, audio_token_index=120)
    for c in (acfg, tcfg, cfg):
        c._attn_implementation = "sdpa"
    assert cfg.use_return_dict and not cfg.output_attentions and not cfg.output_hidden_states

    class AsTuple(torch.nn.Module):                  # 4.51.3 encoder layers return a tuple, 5.x a tensor
        def __init__(self, inner):
            super().__init__()
            self.inner = inner
            self.self_attn = inner.self_attn         # (the reference sets `layer.self_attn.is_causal`, :191-192)

        def forward(self, h, mask=None, **kw):
            return (self.inner(h, mask),)

    class Assembled(torch.nn.Module):                # 4.51.3's attribute layout
        def __init__(self):
            super().__init__()
            self.config = cfg
            self.audio_tower = Qwen2AudioEncoder(acfg)
            self.multi_modal_projector = Qwen2AudioMultiModalProjector(cfg)
            self.language_model = Qwen2ForCausalLM(tcfg)

        def get_input_embeddings(self):
            return self.language_model.get_input_embeddings()
    torch.manual_seed(31)
    model = Assembled().float().train()
    with torch.no_grad():
        model.audio_tower.embed_positions.weight.copy_(torch.randn_like(model.audio_tower.embed_positions.weight) * 0.1)
    _bf16_round_(model)
    model.audio_tower.layers = torch.nn.ModuleList([AsTuple(l) for l in model.audio_tower.layers])
    type(model.audio_tower).forward = ref.forward_audio_tower        # (pre_init :261, on this process's class object)
    assert model.audio_tower.device == torch.device("cpu")

    # a batch as processing_qwen2_audio.py:119-147 yields it: two samples, right-padded; clips of 100 / 61 valid mel frames
    # in a 100-frame buffer -> 25 / 15 audio tokens
    g = torch.Generator().manual_seed(32)
    B, T, Tm = 2, 64, 100
    frames = torch.tensor([100, 61])
    n_tok = (((frames - 1) // 2 + 1) - 2) // 2 + 1
    ids = torch.full((B, T), 0, dtype=torch.int64)
    attn = torch.zeros(B, T, dtype=torch.int64)
    labels = torch.full((B, T), -100, dtype=torch.int64)
    sl = torch.ones(B, T, dtype=torch.int64)
    for b, (na, n_resp) in enumerate(zip(n_tok.tolist(), (11, 7))):
        prompt = torch.cat([torch.randint(3, 110, (5,), generator=g), torch.full((na,), 120), torch.randint(3, 110, (4,), generator=g)])
        resp = torch.randint(3, 110, (n_resp,), generator=g)
        seq = torch.cat([prompt, resp])
        n = seq.numel()
        ids[b, :n], attn[b, :n] = seq, 1
        lab = torch.cat([torch.full((prompt.numel() - 1,), -100), resp, torch.tensor([2])])          # (:100-101, eos = 2)
        labels[b, :n] = lab
        sl[b, :n] = n_resp + 1
    feats = torch.randn(B, 16, Tm, generator=g).bfloat16().float()
    fmask = (torch.arange(Tm)[None, :] < frames[:, None]).to(torch.int64)
    feats = feats * fmask[:, None, :]                                     # the feature extractor pads with zeros
    batch = {"input_ids": ids, "attention_mask": attn, "labels": labels, "sentence_lens": sl, "num_sentence": 2,
             "input_features": feats, "feature_attention_mask": fmask}
    out = ref.forward(model, input_ids=ids, input_features=feats, attention_mask=attn, feature_attention_mask=fmask,
                      shift_labels=labels)
    ps, pt = ref_ce(out.logits, labels, sl, batch["num_sentence"])
    ps.backward()
    strip = lambda n: n.replace(".inner", "")
    arrs = {f"param/{strip(n)}": _bits(p) for n, p in model.named_parameters()}
    arrs.update({f"grad/{strip(n)}": npy(p.grad).astype(np.float16) for n, p in model.named_parameters() if p.grad is not None})
    for k in ("input_ids", "attention_mask", "labels", "sentence_lens", "input_features", "feature_attention_mask"):
        arrs[f"batch/{k}"] = npy(batch[k])
    arrs["batch/num_sentence"] = np.array(2)
    arrs["logits"] = npy(out.logits)
    arrs["loss_per_sample"], arrs["loss_per_token"] = npy(ps), npy(pt)
    arrs["config_json"] = np.array(str({"audio_config": adims, "text_config": tdims, "audio_token_index": 120}))
    save("qwen2_audio_model_dev.npz", **arrs)


def speed_perturb_draws_case():
    """touchnet/data/functions.py:99-114 run here with sox stood in by a recorder (torchaudio is not in this image): WHICH
    speed the reference's stage draws for each of 24 consecutive samples under `random.seed(seed)`, and that it hands
    sox exactly [['speed', s], ['rate', sample_rate]] for s != 1.  The twins replay the draws; the resampling itself has no
    reference output to be held to (libsox), see oracle/frontend.py::speed_perturb."""
    import random
    import torchaudio                                          # the MagicMock installed by _ref_import
    calls = []

    def recorder(waveform, sample_rate, effects):
        calls.append((int(sample_rate), [list(map(str, e)) for e in effects]))
        return waveform, sample_rate
    torchaudio.sox_effects.apply_effects_tensor = recorder
    out = {}
    for name, speeds, seed in (("recipe", [0.9, 1.0, 1.1], 2025), ("two", [0.9, 1.1], 3)):
        cfg = types.SimpleNamespace(audio_speed_perturb_speeds=speeds)
        data = iter([{"sample_rate": 16000, "waveform": torch.zeros(1, 160)} for _ in range(24)])
        calls.clear()
        random.seed(seed)
        n = sum(1 for _ in ref_fn.audio_speed_perturb(data, cfg))
        assert n == 24
        random.seed(seed)
        chosen = [random.choice(speeds) for _ in range(24)]       # (the stage makes exactly this one call per sample)
        assert [float(c[1][0][1]) for c in calls] == [s for s in chosen if s != 1.0]
        assert all(c[1] == [["speed", str(s)], ["rate", "16000"]] for c, s in zip(calls, [s for s in chosen if s != 1.0]))
        out[f"{name}/speeds"] = np.array(speeds)
        out[f"{name}/seed"] = np.array(seed)
        out[f"{name}/chosen"] = np.array(chosen)
    save("speed_perturb_draws.npz", **out)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for fn in (text_cases, asr_cases, unpacked_asr_cases, ce_cases, docmask_cases, rope_cases, tiny_llama_case,
               touch_audio_case, qwen2_audio_tower_case, frontend_cases, fbank_cases, bestrq_cases, touchdataset_case,
               boundary_case, qwen2_audio_data_case, kimi_decoder_case, tiny_llama_dev_case, touch_audio_dev_case,
               qwen2_audio_tower_dev_case, kimi_decoder_dev_case, kimi_audio_input_case,
               kimi_audio_data_case, audiofeat_augment_case, speed_perturb_draws_case, qwen2_audio_model_dev_case):
        if not only or fn.__name__ in only:
            fn()
