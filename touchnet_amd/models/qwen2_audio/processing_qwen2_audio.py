"""Qwen2-Audio SFT batches, PACKED (the reference pads: `dynamic_batch`, touchnet/models/qwen2_audio/
processing_qwen2_audio.py:17-199 — its model path cannot take packed rows; ours can).

Per sample the token-level content is the reference's, line for line in meaning (:41-104):
  prompt   = tokenizer("<|audio_bos|>" + "<|AUDIO|>" * n_audio + "<|audio_eos|>" + instruct)
  n_audio  = ((L - 1) // 2 + 1 - 2) // 2 + 1 with L = valid mel frames (max(3000, frames) when padded to 30 s... see
             `audio_token_count`), input_ids = prompt + response, labels = -100 x (len(prompt) - 1) + response + eos
             (PRE-shifted), sentence_lens = len(response) + 1 on every position of the sample
and the samples are then packed greedily into [B, T] rows like every other packer here (document ids in
`attention_mask`, `position_ids` restarting per sample), with the mel features computed ON THE DEVICE
(tn_log_mel) and handed over as [n_audio_clips, n_mels, frames] plus the flat positions of the AUDIO tokens —
the model scatters the tower output there (index copy instead of the reference's masked_scatter, no host sync).
"""
from __future__ import annotations

import numpy as np
import torch

from touchnet_amd.models.backend import ops
from touchnet_amd.models.llama.processing_llama import PackBuffer

TEMPLATE_S2T = "<|audio_bos|><|AUDIO|><|audio_eos|><|INSTRUCT|>"      # processing_qwen2_audio.py:14
DEFAULT_INSTRUCT = "Generate the transcription:"                        # :43-44
WHISPER_FRAMES = 3000                                                    # 30 s at 10 ms hop
HOP = 160


def audio_token_count(valid_frames: int) -> int:
    """:79-82 — conv stride 2, then average pool 2."""
    return ((valid_frames - 1) // 2 + 1 - 2) // 2 + 1


def _tok_ids(out):
    ids = out["input_ids"] if isinstance(out, dict) or hasattr(out, "keys") else out.input_ids
    if isinstance(ids, torch.Tensor):
        ids = ids.reshape(-1).tolist()
    elif len(ids) and isinstance(ids[0], (list, tuple)):
        ids = list(ids[0])
    return [int(v) for v in ids]


def _sample_tokens(sample, tokenizer, n_audio: int):
    instruct = sample.get("instruct", DEFAULT_INSTRUCT)
    text = TEMPLATE_S2T.replace("<|INSTRUCT|>", instruct).replace("<|AUDIO|>", "<|AUDIO|>" * n_audio, 1)
    prompt = _tok_ids(tokenizer(text, padding=False))
    response = _tok_ids(tokenizer(sample["response"], add_special_tokens=False))
    return prompt, response


def _emit(buf: PackBuffer, segs, mels, valid, pad_id: int, audio_token: int):
    B, T = buf.B, buf.T
    input_ids = np.full(B * T, pad_id, dtype=np.int64)
    labels = np.full(B * T, -100, dtype=np.int64)
    position_ids = np.zeros(B * T, dtype=np.int64)
    doc = np.zeros(B * T, dtype=np.int64)
    sentence_lens = np.ones(B * T, dtype=np.int64)
    audio_positions = np.zeros(0, dtype=np.int64)
    if len(buf):
        seg, within, flat = buf.scatter_index()
        ids = np.concatenate([np.asarray(p + r, dtype=np.int64) for p, r, _ in segs])
        lab = np.concatenate([np.asarray([-100] * (len(p) - 1) + r + [eos], dtype=np.int64) for p, r, eos in segs])
        input_ids[flat], labels[flat] = ids, lab
        position_ids[flat] = within
        doc[flat] = np.asarray(buf.sents, dtype=np.int64)[seg]
        sentence_lens[flat] = np.asarray([len(r) + 1 for _, r, _ in segs], dtype=np.int64)[seg]
        audio_positions = flat[ids == audio_token]
    frames = max(m.shape[0] for m in mels)
    dev = mels[0].device
    feats = torch.zeros(len(mels), mels[0].shape[1], frames, dtype=torch.float32, device=dev)
    for i, m in enumerate(mels):
        feats[i, :, :m.shape[0]] = m.t()
    t = lambda a: torch.from_numpy(a.reshape(B, T))
    lab_t = t(labels)
    return {"input_ids": t(input_ids), "labels": lab_t, "shift_labels": lab_t, "position_ids": t(position_ids),
            "attention_mask": t(doc), "sentence_lens": t(sentence_lens), "num_sentence": len(buf),
            "input_features": feats, "audio_positions": torch.from_numpy(audio_positions),
            "audio_output_lengths": torch.tensor([audio_token_count(v) for v in valid], dtype=torch.int64),
            "labelled_rows_max": int(sum(len(r) + 1 for _, r, _ in segs)),           # response + eos (host int: no sync)
            "valid_rows_max": int((doc > 0).sum())}                                  # non-pad slots (host int: no sync)


def _samples(data, config, tokenizer, n_mels: int, limit: int = None):
    """What both batchers need of a sample, in the reference's order of checks (:37-112): (prompt ids, response ids,
    mel [frames, n_mels] on the device, valid frames).  `limit`: drop samples with more tokens than a row (packed form)."""
    for sample in data:
        if "response" not in sample:
            if "txt" not in sample:
                continue                                                   # :46-50
            sample["response"] = sample["txt"]
        wav = sample["waveform"]
        n = int(wav.shape[-1])
        frames = n // HOP
        # WhisperFeatureExtractor(padding="max_length", truncation=False): zero-pad the WAVEFORM to 30 s, longer audio
        # is kept whole; valid frames = the real ones below 30 s, all of them above (:66-77)
        # (below 30 s HF's frame mask is the SAMPLE mask taken every 160th sample: ceil(n / 160) valid frames)
        total = max(WHISPER_FRAMES, frames)
        L = min(-(-n // HOP), WHISPER_FRAMES) if n <= WHISPER_FRAMES * HOP else total
        if L * 10 > config.audio_max_length_in_ms_for_filter:
            continue                                                       # :84-86
        n_audio = audio_token_count(L)
        prompt, response = _sample_tokens(sample, tokenizer, n_audio)
        tot = len(prompt) + len(response)
        if not (config.text_min_length_in_tokens_for_filter <= tot <= config.text_max_length_in_tokens_for_filter):
            continue                                                       # :106-112
        if limit is not None and tot > limit:
            continue

        def mel(wav=wav, n=n, total=total):
            w = wav.reshape(-1)
            if not w.is_cuda and torch.cuda.is_available():
                w = w.to(torch.device("cuda", torch.cuda.current_device()), non_blocking=True)
            if w.dtype == torch.int16:
                w = ops().pcm16_to_float(w)
            return ops().log_mel_spectrogram(w, n_mels, padding=max(0, WHISPER_FRAMES * HOP - n))[:total]   # [total, n_mels]
        yield prompt, response, mel, L, total


def batch_qwen2_audio_packed(data, config, processor):
    """Datapipe stage `f(data, config, processor)`; `processor` = HF Qwen2AudioProcessor (its `.tokenizer` is used) or
    a bare HF tokenizer.  Audio-token id: `tokenizer.convert_tokens_to_ids("<|AUDIO|>")`."""
    tokenizer = getattr(processor, "tokenizer", processor)
    audio_token = int(tokenizer.convert_tokens_to_ids("<|AUDIO|>"))
    eos, pad = int(tokenizer.eos_token_id), int(tokenizer.pad_token_id)
    n_mels = getattr(config, "audiofeat_num_mel_bins", 128)
    T = config.dataset_text_seqlen
    buf = PackBuffer(config.dataset_batchsize, T)
    segs, mels, valid = [], [], []
    for prompt, response, mel, L, _ in _samples(data, config, tokenizer, n_mels, limit=T):
        tot = len(prompt) + len(response)
        if buf.place(tot):
            yield _emit(buf, segs, mels, valid, pad, audio_token)
            buf.reset()
            segs, mels, valid = [], [], []
            buf.place(tot)
        segs.append((prompt, response, eos))
        mels.append(mel())
        valid.append(L)
    if (not config.dataloader_drop_last_batch) and buf.dirty:
        yield _emit(buf, segs, mels, valid, pad, audio_token)


def _emit_rows(rows, pad: int, eos: int):
    """the reference's padded batch (:119-147): one sample per row, right-padded; features [B, n_mels, frames]"""
    n = max(len(p) + len(r) for p, r, _, _, _ in rows)
    B = len(rows)
    input_ids = np.full((B, n), pad, dtype=np.int64)
    labels = np.full((B, n), -100, dtype=np.int64)
    mask = np.zeros((B, n), dtype=np.int64)
    slen = np.ones((B, n), dtype=np.int64)
    for i, (p, r, _, _, _) in enumerate(rows):
        k = len(p) + len(r)
        input_ids[i, :k] = p + r
        labels[i, :k] = [-100] * (len(p) - 1) + r + [eos]
        mask[i, :k] = 1
        slen[i, :k] = len(r) + 1
    mels = [m() for _, _, m, _, _ in rows]
    frames = max(m.shape[0] for m in mels)
    feats = torch.zeros(B, mels[0].shape[1], frames, dtype=torch.float32, device=mels[0].device)
    fmask = torch.zeros(B, frames, dtype=torch.int64)
    for i, (m, (_, _, _, L, total)) in enumerate(zip(mels, rows)):
        feats[i, :, :m.shape[0]] = m.t()
        fmask[i, :(L if total == WHISPER_FRAMES else total)] = 1          # (:66-72: all ones for audio longer than 30 s)
    lab = torch.from_numpy(labels)
    return {"input_ids": torch.from_numpy(input_ids), "attention_mask": torch.from_numpy(mask), "labels": lab,
            "shift_labels": lab, "input_features": feats, "feature_attention_mask": fmask, "num_sentence": B,
            "sentence_lens": torch.from_numpy(slen),
            "labelled_rows_max": int(sum(len(r) + 1 for _, r, _, _, _ in rows)), "valid_rows_max": int(mask.sum())}


def dynamic_batch(data, config, processor):
    """The reference's OWN (unpacked) batcher, processing_qwen2_audio.py:17-199 — what `--dataset_enable_pack false` gives:
    one sample per row, right-padded to the longest of the batch, a batch closed when (rows + 1) x longest exceeds
    batchsize x seqlen (:114-116; the new sample opens the next one).  Same keys as the reference's batch
    (`feature_attention_mask`, 0 / 1 `attention_mask`); the product model takes them as they are
    (modeling_qwen2_audio.py).  The mel features come from the device kernel."""
    tokenizer = getattr(processor, "tokenizer", processor)
    eos, pad = int(tokenizer.eos_token_id), int(tokenizer.pad_token_id)
    n_mels = getattr(config, "audiofeat_num_mel_bins", 128)
    budget = config.dataset_batchsize * config.dataset_text_seqlen
    rows, longest = [], 0
    for item in _samples(data, config, tokenizer, n_mels):
        tot = len(item[0]) + len(item[1])
        longest = max(longest, tot)
        if longest * (len(rows) + 1) > budget:
            if rows:
                yield _emit_rows(rows, pad, eos)
            rows, longest = [item], tot
        else:
            rows.append(item)
    if (not config.dataloader_drop_last_batch) and rows:
        yield _emit_rows(rows, pad, eos)
