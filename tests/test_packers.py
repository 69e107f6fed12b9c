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
