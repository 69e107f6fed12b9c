"""Synthetic packed batches of the shapes BASELINE.json / SURVEY.md §8d name (no datasets or checkpoints
are reachable: weights are random-init, data is synthetic — and says so in every bench line).

All generators are seeded with the reference's default seed 2025 (touchnet/bin/__init__.py:223-228) plus
the dp rank, and go through the SAME packers the real datapipe ends with.
"""
from __future__ import annotations

import types

import numpy as np
import torch

from touchnet_amd.models.llama.processing_llama import batch_text
from touchnet_amd.models.touch_audio.processing_touch_audio import batch_pairaudio_pairtext_packed


def _tok(vocab):
    return types.SimpleNamespace(bos=1, eos=2, pad=0)


def text_batch(vocab: int, batchsize: int, seqlen: int, seed: int = 2025, min_len: int = 1, max_len: int = 12):
    """Config A: sentences of U{min..max} tokens packed into [B, T]."""
    rng = np.random.RandomState(seed)
    cfg = types.SimpleNamespace(dataset_batchsize=batchsize, dataset_text_seqlen=seqlen,
                                dataloader_drop_last_batch=False)

    def gen():
        while True:
            n = int(rng.randint(min_len, max_len + 1))
            yield {"input_ids": [int(v) for v in rng.randint(3, vocab, size=n)]}
    return next(iter(batch_text(gen(), cfg, _tok(vocab))))


def asr_waveforms(seqlen: int, batchsize: int, seed: int = 2025, frames_per_sec: float = 25.0,
                  min_s: float = 1.5, max_s: float = 14.5, min_tok: int = 4, max_tok: int = 25):
    """Config B: AISHELL-shaped utterances — duration U[1.5, 14.5] s of N(0, 0.1^2) noise clipped to
    [-1, 1] at 16 kHz, transcripts of U{4..25} tokens; enough of them to fill B rows of T slots."""
    rng = np.random.RandomState(seed)
    utts, slots = [], 0
    while slots < batchsize * seqlen * 1.05:
        dur = float(rng.uniform(min_s, max_s))
        n = int(dur * 16000)
        ntok = int(rng.randint(min_tok, max_tok + 1))
        utts.append((n, ntok))
        slots += int(dur * frames_per_sec) + ntok + 1
    return utts, rng


def asr_batch_from_device_frontend(vocab: int, batchsize: int, seqlen: int, device, num_mel_bins: int = 80,
                                   stack: int = 5, stride: int = 4, seed: int = 2025, frontend=None, wavs=None,
                                   utts=None):
    """waveform (on device) -> kaldi fbank -> stack/stride/normalise (HIP kernels) -> packed ASR batch.
    `frontend` = op namespace (touchnet_amd.functional on the GPU).  Returns (batch, wavs, utts) so a
    benchmark can keep the waveforms resident and re-run the frontend inside the timed step."""
    if utts is None:
        utts, rng = asr_waveforms(seqlen, batchsize, seed)
    else:
        rng = np.random.RandomState(seed + 1)
    if wavs is None:
        g = torch.Generator(device="cpu").manual_seed(seed)
        wavs = [(torch.randn(n, generator=g) * 0.1).clamp_(-1, 1).to(device) for n, _ in utts]
    tok_rng = np.random.RandomState(seed + 7)
    cfg = types.SimpleNamespace(dataset_batchsize=batchsize, dataset_text_seqlen=seqlen, dataset_audio_seqlen=seqlen,
                                audiofeat_num_mel_bins=num_mel_bins, audiofeat_stack_length=stack,
                                dataloader_drop_last_batch=False)

    def gen():
        for wav, (_, ntok) in zip(wavs, utts):
            feat = frontend.audiofeat_stack(frontend.kaldi_fbank(wav, num_mel_bins), stack, stride, True)
            yield {"audiofeat": feat, "input_ids": [int(v) for v in tok_rng.randint(3, vocab, size=ntok)]}
    batch = next(iter(batch_pairaudio_pairtext_packed(gen(), cfg, _tok(vocab))))
    return batch, wavs, utts


def qwen2_audio_tokens_of(n_samples: int) -> int:
    """AUDIO placeholder tokens of an utterance of n_samples at 16 kHz (touchnet/models/qwen2_audio/
    processing_qwen2_audio.py:78-82: valid mel frames of WhisperFeatureExtractor's mask -> conv stride 2 -> pool 2)."""
    mel = min(-(-n_samples // 160), 3000)
    inp = (mel - 1) // 2 + 1
    return (inp - 2) // 2 + 1


def qwen2_audio_plan(vocab: int, audio_token: int, batchsize: int, seqlen: int, seed: int = 2025,
                     audio_tokens: int = 750, prompt_pre: int = 8, prompt_post: int = 6, min_resp: int = 5,
                     max_resp: int = 40, audio_seconds=None):
    """Config C (Qwen2-Audio-7B ASR SFT, packed — beyond what the reference can run):
    per sample a 30 s-padded utterance -> 750 AUDIO placeholder tokens inside a ~14-token prompt, then a
    response of U{5..40} tokens + eos.  Labels follow processing_qwen2_audio.py:96-104: -100 on the prompt
    (pre-shifted), response + eos after it; sentence_lens = len(response) + 1 over the whole sample.
    Samples are packed greedily into [B, T] with document ids and per-sample position ids.
    ``audio_seconds=(lo, hi)``: utterances of U[lo, hi] seconds instead — the number of AUDIO tokens then follows the
    valid length like in the reference's processor (a 30 s-padded clip of 8 s gives 200 tokens, not 750) and the
    result carries ``audio_samples`` (valid samples per clip).
    Returns the int64 batch tensors + the number of audio clips (their waveforms are made by the caller)."""
    rng = np.random.RandomState(seed)
    B, T = batchsize, seqlen
    input_ids = np.zeros((B, T), dtype=np.int64)
    labels = np.full((B, T), -100, dtype=np.int64)
    position_ids = np.zeros((B, T), dtype=np.int64)
    doc = np.zeros((B, T), dtype=np.int64)
    sentence_lens = np.ones((B, T), dtype=np.int64)
    audio_pos, n_sent, n_lab = [], 0, 0
    fixed_tokens, lengths, samples = audio_tokens, [], []
    for b in range(B):
        col, d = 0, 1
        while True:
            nresp = int(rng.randint(min_resp, max_resp + 1))
            n_samp = 480000
            if audio_seconds is not None:
                n_samp = int(rng.uniform(*audio_seconds) * 16000)
                audio_tokens = qwen2_audio_tokens_of(n_samp)
            else:
                audio_tokens = fixed_tokens
            plen = prompt_pre + audio_tokens + prompt_post
            tot = plen + nresp
            if col + tot > T:
                break
            ids = np.concatenate([rng.randint(3, vocab - 2000, size=prompt_pre),
                                  np.full(audio_tokens, audio_token),
                                  rng.randint(3, vocab - 2000, size=prompt_post),
                                  rng.randint(3, vocab - 2000, size=nresp)])
            input_ids[b, col:col + tot] = ids
            labels[b, col + plen - 1:col + tot - 1] = ids[plen:]
            labels[b, col + tot - 1] = 2                                   # eos
            position_ids[b, col:col + tot] = np.arange(tot)
            doc[b, col:col + tot] = d
            sentence_lens[b, col:col + tot] = nresp + 1
            audio_pos.append(b * T + col + prompt_pre + np.arange(audio_tokens))
            lengths.append(audio_tokens)
            samples.append(n_samp)
            col += tot
            d += 1
            n_sent += 1
            n_lab += nresp + 1
    t = torch.from_numpy
    return {"input_ids": t(input_ids), "labels": t(labels), "position_ids": t(position_ids),
            "attention_mask": t(doc), "sentence_lens": t(sentence_lens), "num_sentence": n_sent,
            "audio_positions": t(np.concatenate(audio_pos)),
            "audio_output_lengths": torch.tensor(lengths, dtype=torch.int64),
            "audio_samples": samples, "labelled_rows_max": n_lab, "valid_rows_max": int((doc > 0).sum())}, n_sent


def qwen2_audio_long_plan(vocab: int, audio_token: int, batchsize: int, seqlen: int, seed: int = 2025,
                          clips_per_doc=(40, 20, 10, 4, 1), tokens_per_clip: int = 750, prompt_pre: int = 8,
                          prompt_post: int = 6, resp_per_clip=(40, 60)):
    """Config D ("one 20-min recording per row ... 30 000 audio tokens x 2 docs + text to fill
    T=65536"): a long recording reaches Qwen2-Audio as consecutive 30 s windows (the tower's `max_source_positions`
    1500 frames -> 750 AUDIO tokens each), so a document = prompt + K x 750 AUDIO tokens + a transcript of
    K x U{resp_per_clip} tokens + eos.  Rows are filled greedily with the largest K of `clips_per_doc` that still fits
    (40 clips = 20 min first).  Label / sentence_lens / position conventions are `qwen2_audio_plan`'s.
    Returns the batch tensors (+ `labelled_rows_max`) and the number of 30 s clips (waveforms are the caller's)."""
    rng = np.random.RandomState(seed)
    B, T = batchsize, seqlen
    top = vocab - 2000 if vocab > 4000 else vocab                       # (plain text ids; tiny test vocabularies: all of it)
    input_ids = np.zeros((B, T), dtype=np.int64)
    labels = np.full((B, T), -100, dtype=np.int64)
    position_ids = np.zeros((B, T), dtype=np.int64)
    doc = np.zeros((B, T), dtype=np.int64)
    sentence_lens = np.ones((B, T), dtype=np.int64)
    audio_pos, lengths, n_sent, n_lab = [], [], 0, 0
    for b in range(B):
        col, d = 0, 1
        while True:
            pick = None
            for K in clips_per_doc:
                nresp = int(sum(rng.randint(resp_per_clip[0], resp_per_clip[1] + 1) for _ in range(K)))
                plen = prompt_pre + K * tokens_per_clip + prompt_post
                if col + plen + nresp <= T:
                    pick = (K, nresp, plen)
                    break
            if pick is None:
                break
            K, nresp, plen = pick
            tot = plen + nresp
            ids = np.concatenate([rng.randint(3, top, size=prompt_pre),
                                  np.full(K * tokens_per_clip, audio_token),
                                  rng.randint(3, top, size=prompt_post),
                                  rng.randint(3, top, size=nresp)])
            input_ids[b, col:col + tot] = ids
            labels[b, col + plen - 1:col + tot - 1] = ids[plen:]
            labels[b, col + tot - 1] = 2
            position_ids[b, col:col + tot] = np.arange(tot)
            doc[b, col:col + tot] = d
            sentence_lens[b, col:col + tot] = nresp + 1
            audio_pos.append(b * T + col + prompt_pre + np.arange(K * tokens_per_clip))
            lengths += [tokens_per_clip] * K
            col += tot
            d += 1
            n_sent += 1
            n_lab += nresp + 1
    t = torch.from_numpy
    return {"input_ids": t(input_ids), "labels": t(labels), "position_ids": t(position_ids),
            "attention_mask": t(doc), "sentence_lens": t(sentence_lens), "num_sentence": n_sent,
            "audio_positions": t(np.concatenate(audio_pos)),
            "audio_output_lengths": torch.tensor(lengths, dtype=torch.int64),
            "labelled_rows_max": n_lab, "valid_rows_max": int((doc > 0).sum())}, len(lengths)


def kimi_audio_plan(vocab_text: int, audio_code_base: int, n_codes: int, batchsize: int, seqlen: int, seed: int = 2025,
                    blank_id: int = 0, audio_s=(2.0, 14.5), codes_per_s: float = 12.5, text_tokens=(5, 40),
                    media_markers=None):
    """Config E ("text / audio id streams of equal length (processing_kimi_audio.py:112-116),
    V=168448 logits, text head only"): interleaved audio/text documents — a span of discrete audio codes (12.5 Hz GLM-4
    voice tokens of a U[audio_s] s utterance) on the AUDIO stream with blanks on the TEXT stream, then its transcript on
    the text stream with blanks on the audio stream; the text head is trained on the transcript (labels pre-shifted,
    per-sentence normalisation like the other recipes).  Documents are packed greedily into [B, T]."""
    rng = np.random.RandomState(seed)
    B, T = batchsize, seqlen
    text = np.full((B, T), blank_id, dtype=np.int64)
    audio = np.full((B, T), blank_id, dtype=np.int64)
    labels = np.full((B, T), -100, dtype=np.int64)
    position_ids = np.zeros((B, T), dtype=np.int64)
    doc = np.zeros((B, T), dtype=np.int64)
    sentence_lens = np.ones((B, T), dtype=np.int64)
    n_sent = n_lab = 0
    clip_codes = []
    for b in range(B):
        col, d = 0, 1
        while True:
            na = int(rng.uniform(*audio_s) * codes_per_s)
            nt = int(rng.randint(text_tokens[0], text_tokens[1] + 1))
            tot = 2 + na + nt                                            # <media_begin> codes <media_end> transcript
            if col + tot > T:
                break
            a0 = col + 1
            codes = rng.randint(0, n_codes, size=na)
            if media_markers is None:
                audio[b, a0:a0 + na] = audio_code_base + codes
                text[b, col], text[b, a0 + na] = 3, 4
            else:
                # the reference's prompt form (processing_kimi_audio.py:33-34): blanks between <|im_media_begin|> and
                # <|im_media_end|> on the AUDIO stream — the model writes the continuous speech embeddings there
                # (modeling_kimi_audio.py:969-978); the speech tokenizer's ids of the clip travel beside the batch
                audio[b, col], audio[b, a0 + na] = media_markers
                clip_codes.append(codes)
            ids = rng.randint(5, vocab_text - 2000, size=nt)
            t0 = a0 + na + 1
            text[b, t0:t0 + nt] = ids
            labels[b, t0 - 1:t0 + nt - 1] = ids
            labels[b, t0 + nt - 1] = 2
            position_ids[b, col:col + tot] = np.arange(tot)
            doc[b, col:col + tot] = d
            sentence_lens[b, col:col + tot] = nt + 1
            col += tot
            d += 1
            n_sent += 1
            n_lab += nt + 1
    t = torch.from_numpy
    out = {"text_input_ids": t(text), "audio_input_ids": t(audio), "labels": t(labels),
           "position_ids": t(position_ids), "attention_mask": t(doc), "sentence_lens": t(sentence_lens),
           "num_sentence": n_sent, "labelled_rows_max": n_lab}
    if media_markers is not None:
        ids = np.zeros((len(clip_codes), 375), dtype=np.int64)         # 30 s-padded clips: 375 tokens, the first na used
        for i, c in enumerate(clip_codes):
            ids[i, :len(c)] = c
        out["speech_tokenizer_ids"] = t(ids)
        out["clip_tokens"] = [len(c) for c in clip_codes]
    return out
