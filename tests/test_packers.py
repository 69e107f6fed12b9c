"""Product packers (touchnet_amd.models.*.processing_*) against the reference-generated goldens: bit exact."""
import types

import numpy as np
import pytest
import torch

from touchnet_amd.models.llama.processing_llama import batch_text
from touchnet_amd.models.touch_audio.processing_touch_audio import batch_pairaudio_pairtext_packed

TOK = types.SimpleNamespace(bos=1, eos=2, pad=0)


def _split(flat, lens):
    out, o = [], 0
    for n in lens:
        out.append([int(v) for v in flat[o:o + n]])
        o += n
    return out


@pytest.mark.parametrize("case", ["overflow", "exactfit", "droplast", "single"])
def test_batch_text(golden, case):
    g = golden("packing_text.npz")
    B, T, drop, nb = [int(v) for v in g[f"{case}/meta"]]
    sents = _split(g[f"{case}/tokens"], g[f"{case}/lens"])
    cfg = types.SimpleNamespace(dataset_batchsize=B, dataset_text_seqlen=T, dataloader_drop_last_batch=bool(drop))
    got = list(batch_text(({"input_ids": s} for s in sents), cfg, TOK))
    assert len(got) == nb
    for i, b in enumerate(got):
        for k in ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens"):
            assert b[k].dtype == torch.int64
            np.testing.assert_array_equal(b[k].numpy(), g[f"{case}/b{i}/{k}"], err_msg=f"{case} b{i} {k}")
        assert b["num_sentence"] == int(g[f"{case}/b{i}/num_sentence"]) and b["inputs_embeds"] is None


@pytest.mark.parametrize("case", ["mixed", "droplast"])
def test_batch_asr(golden, case):
    g = golden("packing_asr.npz")
    B, T, drop, nb, F = [int(v) for v in g[f"{case}/meta"]]
    ids = _split(g[f"{case}/tokens"], g[f"{case}/tlens"])
    feats, o = [], 0
    for a in g[f"{case}/alens"]:
        feats.append(torch.from_numpy(g[f"{case}/feats"][o:o + a]))
        o += a
    cfg = types.SimpleNamespace(dataset_batchsize=B, dataset_text_seqlen=T, dataset_audio_seqlen=T,
                                audiofeat_num_mel_bins=F, audiofeat_stack_length=1,
                                dataloader_drop_last_batch=bool(drop))
    data = ({"audiofeat": f, "input_ids": i} for f, i in zip(feats, ids))
    got = list(batch_pairaudio_pairtext_packed(data, cfg, TOK))
    assert len(got) == nb
    for i, b in enumerate(got):
        for k in ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens", "input_features",
                  "shift_labels"):
            np.testing.assert_array_equal(b[k].numpy(), g[f"{case}/b{i}/{k}"], err_msg=f"{case} b{i} {k}")
        assert b["num_sentence"] == int(g[f"{case}/b{i}/num_sentence"])
        assert b["shift_labels"] is b["labels"]


@pytest.mark.parametrize("case", ["mixed", "droplast", "tight"])
def test_unpacked_touch_audio_batchers_equal_the_reference(golden, case):
    """`batch_pairaudio_pairtext` / `batch_audio` (processing_touch_audio.py:217-428, `--dataset_enable_pack false`): batch
    boundaries (the flush rule with its running maximum, which a skipped over-long sample still moves), padding values and
    every tensor equal the reference run; keys the reference leaves `None` are `None`."""
    import types
    from touchnet_amd.models.touch_audio.processing_touch_audio import batch_audio, batch_pairaudio_pairtext
    g = golden("unpacked_asr.npz")
    B, T, drop, F = [int(v) for v in g[f"{case}/meta"]]
    alens, tlens = g[f"{case}/alens"], g[f"{case}/tlens"]
    feats = np.split(g[f"{case}/feats"], np.cumsum(alens)[:-1])
    ids = [[int(v) for v in t] for t in np.split(g[f"{case}/tokens"], np.cumsum(tlens)[:-1])]
    cfg = types.SimpleNamespace(dataset_batchsize=B, dataset_text_seqlen=T, dataset_audio_seqlen=T,
                                dataloader_drop_last_batch=bool(drop))

    class Codes:
        def tokenize(self, feat):
            return [int(v) for v in (feat.sum(1) * 7.0).abs().long() % 50]
    for kind, fn, tok in (("pairs", batch_pairaudio_pairtext, TOK), ("audio", batch_audio, Codes())):
        data = ({"audiofeat": torch.from_numpy(f.copy()), "input_ids": i} for f, i in zip(feats, ids))
        batches = list(fn(data, cfg, tok))
        assert len(batches) == int(g[f"{case}/{kind}/n"]) and len(batches) >= 1, (kind, len(batches))
        for i, b in enumerate(batches):
            for k in ("input_ids", "input_features", "labels", "shift_labels", "position_ids", "attention_mask",
                      "sentence_lens", "num_sentence"):
                key = f"{case}/{kind}/b{i}/{k}"
                if key + "/none" in g.files:
                    assert b[k] is None, (kind, i, k)
                elif k == "num_sentence":
                    assert int(b[k]) == int(g[key])
                else:
                    got = b[k].numpy()
                    assert got.dtype == g[key].dtype and np.array_equal(got, g[key]), (kind, i, k)


def test_empty_input_yields_nothing():
    cfg = types.SimpleNamespace(dataset_batchsize=2, dataset_text_seqlen=8, dataloader_drop_last_batch=False)
    assert list(batch_text(iter([]), cfg, TOK)) == []


@pytest.mark.parametrize("case", ["overflow", "droplast", "tail"])
def test_batch_audio_packed(golden, case):
    """Product audio-pretrain packer vs the reference-generated buffers (bit exact).  The tokenizer here is the
    oracle's (CPU); the HIP tokenizer is checked against the same goldens in tests/test_kernels_gpu.py."""
    from oracle import tokenizer as otok
    from touchnet_amd.models.touch_audio import batch_audio_packed
    g = golden("bestrq.npz")
    B, T, drop = [int(v) for v in g[f"pack/{case}/cfg"]]
    q, c = otok.bestrq_tables(64, 24, 8, 7)
    tok = types.SimpleNamespace(tokenize=lambda f: otok.bestrq_tokenize(f.numpy(), q, c).tolist())
    feats, o = [], 0
    for n in g[f"pack/{case}/lens"]:
        feats.append(torch.from_numpy(g[f"pack/{case}/feats"][o:o + n]))
        o += n
    cfg = types.SimpleNamespace(dataset_batchsize=B, dataset_audio_seqlen=T, audiofeat_num_mel_bins=6,
                                audiofeat_stack_length=4, dataloader_drop_last_batch=bool(drop))
    got = list(batch_audio_packed(({"audiofeat": f} for f in feats), cfg, tok))
    assert len(got) == int(g[f"pack/{case}/n"])
    for i, b in enumerate(got):
        assert b["input_ids"] is None and b["shift_labels"] is b["labels"]
        for k in ("input_features", "labels", "position_ids", "attention_mask", "sentence_lens"):
            assert b[k].dtype == (torch.float32 if k == "input_features" else torch.int64)
            np.testing.assert_array_equal(b[k].numpy(), g[f"pack/{case}/{i}/{k}"], err_msg=f"{case} b{i} {k}")
        assert b["num_sentence"] == int(g[f"pack/{case}/{i}/num_sentence"])


# ------------------------------------------------------------------ randomised: product packers == oracle restatement
def _eq_batches(got, want, keys):
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        for k in keys:
            av = a[k].numpy() if isinstance(a[k], torch.Tensor) else a[k]
            np.testing.assert_array_equal(av, b[k], err_msg=f"batch {i} {k}")
        assert a["num_sentence"] == b["num_sentence"]


@pytest.mark.parametrize("seed", range(12))
def test_packers_equal_oracle_on_random_streams(seed):
    """The vectorised product packers against the loop restatements of oracle/ (which the reference goldens pin) on
    random streams: random B, T, drop_last, sentence / utterance lengths incl. ones that exactly fill a row, over-long
    utterances (dropped by the audio packers) and streams that end on a full buffer."""
    from oracle import packing as opk
    from oracle import tokenizer as otok
    from touchnet_amd.models.touch_audio import batch_audio_packed
    rng = np.random.RandomState(100 + seed)
    B, T = int(rng.randint(1, 5)), int(rng.choice([8, 17, 32, 64]))
    drop = bool(rng.randint(2))
    n = int(rng.randint(0, 40))
    # text: sentence slots = len + 1 <= T (over-long text is filtered upstream of the packer)
    sents = [[int(v) for v in rng.randint(3, 50, size=int(rng.randint(0, T)))] for _ in range(n)]
    cfg = types.SimpleNamespace(dataset_batchsize=B, dataset_text_seqlen=T, dataset_audio_seqlen=T,
                                audiofeat_num_mel_bins=3, audiofeat_stack_length=2, dataloader_drop_last_batch=drop)
    ikeys = ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens")
    _eq_batches(list(batch_text(({"input_ids": s} for s in sents), cfg, TOK)),
                list(opk.batch_text(sents, B, T, TOK.bos, TOK.eos, TOK.pad, drop_last=drop)), ikeys)
    # ASR pairs: audio frames + text, some segments longer than a row
    pairs = []
    for _ in range(n):
        ta = int(rng.randint(1, T + 3))
        pairs.append((rng.randn(ta, 6).astype(np.float32), [int(v) for v in rng.randint(3, 50, size=int(rng.randint(0, 6)))]))
    got = list(batch_pairaudio_pairtext_packed(({"audiofeat": torch.from_numpy(f), "input_ids": s} for f, s in pairs),
                                               cfg, TOK))
    want = list(opk.batch_pairaudio_pairtext_packed(pairs, B, T, 6, TOK.bos, TOK.eos, TOK.pad, drop_last=drop))
    _eq_batches(got, want, ikeys + ("input_features",))
    # audio pretraining: frames only, labels from a (deterministic stand-in) tokenizer
    tok = types.SimpleNamespace(tokenize=lambda f: [int(v) for v in (np.asarray(f)[:, 0] * 7).astype(np.int64) % 11])
    feats = [f for f, _ in pairs]
    got = list(batch_audio_packed(({"audiofeat": torch.from_numpy(f)} for f in feats), cfg, tok))
    want = list(otok.batch_audio_packed(feats, B, T, 6, tok.tokenize, drop_last=drop))
    _eq_batches(got, want, ("labels", "position_ids", "attention_mask", "sentence_lens", "input_features"))


def test_synthetic_short_utterance_plan_counts_audio_tokens_like_the_batcher():
    """bench.py --workload qwen2_audio_7b_short: the plan's AUDIO-token count per utterance must be what the product
    batcher (pinned against the reference's dynamic_batch in qwen2_audio_data.npz) emits for a clip of that many samples,
    and the packed batch must be consistent with it."""
    import numpy as np
    from touchnet_amd.data import synthetic
    from touchnet_amd.models.qwen2_audio.processing_qwen2_audio import audio_token_count
    for n in (1, 159, 160, 161, 16000, 40051, 231999, 479999, 480000):
        assert synthetic.qwen2_audio_tokens_of(n) == audio_token_count(min(-(-n // 160), 3000))
    tok, n_clips = synthetic.qwen2_audio_plan(156032, 151646, 2, 8192, 7, audio_seconds=(2.0, 14.5))
    lens = tok["audio_output_lengths"].numpy()
    assert len(lens) == n_clips == len(tok["audio_samples"]) and n_clips > 40          # ~32 utterances per 8192-row
    assert [synthetic.qwen2_audio_tokens_of(s) for s in tok["audio_samples"]] == lens.tolist()
    assert tok["audio_positions"].numel() == int(lens.sum())
    ids = tok["input_ids"].reshape(-1).numpy()
    assert (ids[tok["audio_positions"].numpy()] == 151646).all() and (ids == 151646).sum() == lens.sum()
    # the default plan (the metric's workload) is untouched by the new option
    ref, n_ref = synthetic.qwen2_audio_plan(156032, 151646, 2, 8192, 2025)
    assert n_ref == 20 and ref["audio_positions"].numel() == 15000 and ref["labelled_rows_max"] == 526
    assert np.unique(ref["audio_output_lengths"].numpy()).tolist() == [750]


class _KimiTok:
    """the stand-in tokenizer tests/golden/make_golden.py::_KimiTokenizer ran the reference's batcher with"""
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


def test_kimi_audio_batcher_equals_the_reference_dynamic_batch(golden):
    """models/kimi_audio/processing_kimi_audio.py::batch_kimi_audio against the batches the reference's `dynamic_batch`
    (touchnet/models/kimi_audio/processing_kimi_audio.py:37-224) produced from the same waveforms (kimi_audio_data.npz):
    both token streams, labels, sentence lengths, masks and the batching rule bit-exact; log-mel features (oracle
    frontend here, the HIP kernel is held to the same fixture family on the device) at 1e-3."""
    import types

    import numpy as np
    import oracle.ops as oops
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.kimi_audio.processing_kimi_audio import batch_kimi_audio, num_audio_tokens
    g = golden("kimi_audio_data.npz")
    rng = np.random.RandomState(int(g["wave_seed"]))
    samples = []
    for d, tx in zip(g["durations"], g["texts"]):
        n = int(float(d) * 16000)
        samples.append({"waveform": torch.from_numpy((rng.randn(1, n) * 0.05).astype(np.float32)), "txt": str(tx)})
    samples[2]["instruct"] = "Translate:"
    cfg = types.SimpleNamespace(dataset_batchsize=2, dataset_text_seqlen=100, dataloader_drop_last_batch=False,
                                text_min_length_in_tokens_for_filter=1, text_max_length_in_tokens_for_filter=420)
    with use_ops(oops):
        batches = list(batch_kimi_audio(iter(samples), cfg, None, _KimiTok(),
                                        speech_tokenizer=lambda f, m: torch.arange(375) % 7))
    assert len(batches) == int(g["n_batches"]) and len(batches) >= 3
    for i, b in enumerate(batches):
        for k in ("text_input_ids", "audio_input_ids", "attention_mask", "labels", "sentence_lens"):
            assert torch.equal(b[k], torch.tensor(g[f"b{i}/{k}"])), (i, k)
        assert b["num_sentence"] == int(g[f"b{i}/num_sentence"])
        assert torch.equal(b["whisper_attention_mask"].sum(1).cpu().long(), torch.tensor(g[f"b{i}/whisper_attention_mask_sum"]).long())
        feat = b["whisper_input_features"].cpu().numpy()
        assert feat.shape[1:] == (128, 3000)
        np.testing.assert_allclose(feat[:, :, :40], g[f"b{i}/mel_head"], atol=1e-3)
        np.testing.assert_allclose(feat[:, :, ::97], g[f"b{i}/mel_strided"], atol=1e-3)
        # one media-marker pair per row with exactly num_audio_tokens blanks between them; ids for the model's scatter
        a = b["audio_input_ids"]
        for r in range(a.shape[0]):
            p0, p1 = int((a[r] == 302).nonzero()[0]), int((a[r] == 303).nonzero()[0])
            assert p1 - p0 - 1 == -(-int(b["whisper_attention_mask"][r].sum()) // 8)
        assert b["speech_tokenizer_ids"].shape == (a.shape[0], 375)
        assert b["labelled_rows_max"] == int((b["labels"] != -100).sum())
    assert num_audio_tokens(1) == 1 and num_audio_tokens(1281) == 2 and num_audio_tokens(480000) == 375
