"""Kimi-Audio SFT batches — twin of `dynamic_batch` (touchnet/models/kimi_audio/processing_kimi_audio.py:37-224).

Per sample, token for token what the reference builds (:61-126):
  features  WhisperFeatureExtractor(padding="max_length"): the waveform zero-padded to 30 s -> log-mel [128, 3000] and the
            frame mask; `num_audio_tokens = mask[::2][::4].sum()` (conv stride 2, x4 stack: one token per 80 ms)
  text      TEXT template  = user_msg_start, instruct, text_blank, text_blank x n_audio, text_blank x 4
  audio     AUDIO template = text_blank, text_blank x len(instruct), media_begin, text_blank x n_audio, media_end,
            speech_ct_id, msg_end, assistant_msg_start
            (the two streams are tokenised from these strings and must come out equally long, :112-113)
  response  text stream: the response tokens; audio stream: text_blank x len(response)
  labels    -100 x (len(prompt) - 1) + response + text_eos (PRE-shifted), sentence_lens = len(response) + 1 everywhere
and the reference's batching rule: samples are collected until `longest x (n + 1)` would exceed
`dataset_batchsize x dataset_text_seqlen`, then right-padded per row (pad id / 0 / -100 / 1) — one sample, i.e. one media
marker pair, per row (`create_mask_between_markers` relies on it).  The log-mel runs on the DEVICE (tn_log_mel).

Beyond the reference's keys the batch may carry `speech_tokenizer_ids` [n, 375]: the reference model runs the frozen
GLM-4-voice tokenizer inside its forward (modeling_kimi_audio.py:957-963); here that is the loader's job — pass
`speech_tokenizer=callable(features [128, 3000], mask [3000]) -> ids [375]` — and the model takes the ids as an input.
"""
from __future__ import annotations

import torch

from touchnet_amd.models.backend import ops

TEXT_TEMPLATE_S2T = ("<|im_kimia_user_msg_start|><|INSTRUCT|><|im_kimia_text_blank|><|AUDIO|><|im_kimia_text_blank|>"
                     "<|im_kimia_text_blank|><|im_kimia_text_blank|><|im_kimia_text_blank|>")                    # :33
AUDIO_TEMPLATE_S2T = ("<|im_kimia_text_blank|><|INSTRUCT|><|im_media_begin|><|AUDIO|><|im_media_end|>"
                      "<|im_kimia_speech_ct_id|><|im_msg_end|><|im_kimia_assistant_msg_start|>")                  # :34
BLANK, EOS = "<|im_kimia_text_blank|>", "<|im_kimia_text_eos|>"
DEFAULT_INSTRUCT = "Generate the transcription:"                                                                  # :64
WHISPER_FRAMES, HOP, N_MELS = 3000, 160, 128


def num_audio_tokens(n_samples: int) -> int:
    """:81 — valid frames L = ceil(n / 160) capped at 3000 (HF takes the sample mask every 160th sample); one token per 8
    frames whose first frame is valid."""
    L = min(-(-n_samples // HOP), WHISPER_FRAMES)
    return -(-L // 8)


def _ids(tokenizer, text):
    return [int(v) for v in tokenizer.tokenize(text, add_special_tokens=False)]


def sample_tokens(sample, tokenizer, n_audio: int):
    """-> (text_input_ids, audio_input_ids, labels, sentence_len) as python lists (:83-126)"""
    instruct = sample.get("instruct", DEFAULT_INSTRUCT)
    instruct_ids = _ids(tokenizer, instruct)
    response_ids = _ids(tokenizer, sample["response"])
    text_prompt = TEXT_TEMPLATE_S2T.replace("<|INSTRUCT|>", instruct).replace("<|AUDIO|>", BLANK * n_audio)
    audio_prompt = AUDIO_TEMPLATE_S2T.replace("<|INSTRUCT|>", BLANK * len(instruct_ids)).replace("<|AUDIO|>", BLANK * n_audio)
    tp, ap = _ids(tokenizer, text_prompt), _ids(tokenizer, audio_prompt)
    ar = _ids(tokenizer, BLANK * len(response_ids))
    eos = _ids(tokenizer, EOS)
    if len(tp) != len(ap) or len(ar) != len(response_ids):
        raise ValueError(f"text / audio streams differ in length: {len(tp)} vs {len(ap)}, {len(response_ids)} vs {len(ar)}")
    labels = [-100] * (len(tp) - 1) + response_ids + eos
    return tp + response_ids, ap + ar, labels, len(response_ids) + 1


def _pad(rows, value, dtype=torch.int64):
    n = max(len(r) for r in rows)
    out = torch.full((len(rows), n), value, dtype=dtype)
    for i, r in enumerate(rows):
        out[i, :len(r)] = torch.as_tensor(r, dtype=dtype)
    return out


def _emit(buf, pad_id):
    feats = torch.stack([b["features"] for b in buf])
    out = {"text_input_ids": _pad([b["text"] for b in buf], pad_id),
           "audio_input_ids": _pad([b["audio"] for b in buf], pad_id),
           "attention_mask": _pad([[1] * len(b["labels"]) for b in buf], 0),
           "labels": _pad([b["labels"] for b in buf], -100),
           "whisper_input_features": feats,
           "whisper_attention_mask": torch.stack([b["mask"] for b in buf]),
           "num_sentence": len(buf),
           "sentence_lens": _pad([[b["slen"]] * len(b["labels"]) for b in buf], 1)}
    out["position_ids"] = torch.arange(out["labels"].shape[1]).expand_as(out["labels"]).contiguous()
    out["labelled_rows_max"] = int(sum(b["slen"] for b in buf))              # response + eos rows (host int, no sync)
    if all(b.get("speech_ids") is not None for b in buf):
        out["speech_tokenizer_ids"] = torch.stack([b["speech_ids"] for b in buf])
    return out


def batch_kimi_audio(data, config, processor, tokenizer, speech_tokenizer=None):
    """Datapipe stage with the reference's signature `f(data, config, processor, tokenizer)` (:37-41).  `processor` is only
    consulted for its sampling rate (the features are computed here, on the device)."""
    buf, longest = [], 0
    for sample in data:
        sample = dict(sample)
        if "response" not in sample:
            if "txt" not in sample:
                raise KeyError("sample needs `response` or `txt`")                # (:66-68 asserts)
            sample["response"] = sample["txt"]
        wav = sample["waveform"].reshape(-1)
        n = int(wav.shape[0])
        if not wav.is_cuda and torch.cuda.is_available():
            wav = wav.to(torch.device("cuda", torch.cuda.current_device()), non_blocking=True)
        if wav.dtype == torch.int16:
            wav = ops().pcm16_to_float(wav)
        wav = wav[:WHISPER_FRAMES * HOP]                                          # (the extractor truncates at 30 s)
        mel = ops().log_mel_spectrogram(wav, N_MELS, padding=max(0, WHISPER_FRAMES * HOP - n))[:WHISPER_FRAMES]
        L = min(-(-n // HOP), WHISPER_FRAMES)
        mask = (torch.arange(WHISPER_FRAMES) < L).to(torch.int32)
        n_audio = num_audio_tokens(n)
        text, audio, labels, slen = sample_tokens(sample, tokenizer, n_audio)
        length = len(text)
        if length < config.text_min_length_in_tokens_for_filter or length > config.text_max_length_in_tokens_for_filter:
            continue                                                              # :128-132
        item = {"text": text, "audio": audio, "labels": labels, "slen": slen, "features": mel.t().contiguous(),
                "mask": mask.to(mel.device)}
        if speech_tokenizer is not None:
            item["speech_ids"] = speech_tokenizer(item["features"], item["mask"])
        longest = max(longest, length)
        if longest * (len(buf) + 1) > config.dataset_batchsize * config.dataset_text_seqlen:     # :134-136
            yield _emit(buf, int(tokenizer.pad))
            buf, longest = [item], length
        else:
            buf.append(item)
    if (not config.dataloader_drop_last_batch) and buf:
        yield _emit(buf, int(tokenizer.pad))
