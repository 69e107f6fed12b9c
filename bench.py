#!/usr/bin/env python
"""Headline benchmark of the MI355X packed-sequence training path (contract: see the task statement).

    python bench.py --gpus 1 --steps K --warmup W                      # one MI355X
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W      # N MI355X, FSDP2 over RCCL/xGMI

One "step" = one full training step on one packed batch per GPU (weak scaling): device-side audio
frontend (waveform -> log-mel / fbank) -> model forward -> fused lm_head + packed CE -> backward ->
global-norm clip + AdamW.  Nothing is skipped or cached inside the timed region; inputs (waveforms, token
buffers) are resident in HBM before it starts.

Workloads (BASELINE.json `configs`):
    qwen2_audio_7b  (default) Qwen2-Audio-7B ASR SFT, packed B=2 x T=8192 per GPU   <- the metric's config
    qwen2_audio_7b_short      same model and packing, utterances of U[2, 14.5] s whose AUDIO-token count follows the valid
                              length like in the reference's processor (64 clips per GPU instead of 20): shows the audio
                              tower running on the kept frames only (TN_TOWER_VALID_FRAMES=0: the reference's schedule)
    llama_asr_1b              LlamaForASR-1B, fbank-80 stack5/stride4, packed B=1 x T=8192
    qwen2_audio_7b_long       BASELINE config D: Qwen2-Audio-7B on 20-minute recordings, packed B=1 x T=65536, context
                              parallel (`--cp 4`: zig-zag chunks, K/V halo exchange over xGMI, FSDP2 over dp x cp)
    kimi_audio_7b             BASELINE config E: Kimi-Audio-7B decoder (28 + 6 layers, 168448-way text head), interleaved
                              audio-code / text documents, packed B=2 x T=8192, tensor parallel (`--tp 2`) x FSDP2
    tiny                      2-layer d=256 smoke configuration
Mesh flags: `--cp N` / `--tp N` split the `--gpus` ranks as dp x cp x tp (dp = gpus / (cp * tp); mesh and process groups are
the reference's ParallelDims, touchnet/utils/distributed.py:71-196).  `--emulate-rank r` (with --gpus 1) runs rank r of the
cp- or tp-way group ALONE on one MI355X: its shards, its kernels, its share of the clips and tokens; the exchanges with the
peers are skipped (context parallelism: the K/V chunks a real run would receive are stand-in copies of local chunks, so the
attention kernels do the same tile work on other values).  Such a line is the per-GPU compute time of that rank — an upper
bound on what the group can reach — and says so in `config.emulated`.
Weights are random-init (HF init, seed 2025), data is synthetic: there is no network for checkpoints/datasets.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK = 2.5e15   # dense bf16 FLOP/s per MI355X (MI355X_MICROARCH.md; vendor 5 PF figure is 2:1 sparse)
HBM_PEAK = 8.0e12


def qwen2_audio_7b_config():
    from touchnet_amd.models.qwen2_audio import Qwen2AudioConfig
    # examples/audio/sft/asr/wenetspeech/config/Qwen2-Audio-7B.json (shape source)
    return Qwen2AudioConfig.from_dict({
        "audio_config": {"d_model": 1280, "encoder_attention_heads": 20, "encoder_ffn_dim": 5120,
                         "encoder_layers": 32, "max_source_positions": 1500, "num_mel_bins": 128,
                         "init_std": 0.02},
        "audio_token_index": 151646,
        "text_config": {"model_type": "qwen2", "hidden_size": 4096, "intermediate_size": 11008,
                        "num_attention_heads": 32, "num_hidden_layers": 32, "num_key_value_heads": 32,
                        "rms_norm_eps": 1e-5, "rope_theta": 10000, "vocab_size": 156032,
                        "initializer_range": 0.02, "tie_word_embeddings": False}})


def llama_1b_text_config():
    from touchnet_amd.models.llama import DecoderConfig
    # examples/text/pretrain/fineweb-edu/config/Llama-3_2-1B.json
    return DecoderConfig.from_dict({
        "model_type": "llama", "hidden_size": 2048, "intermediate_size": 8192, "num_attention_heads": 32,
        "num_hidden_layers": 16, "num_key_value_heads": 8, "head_dim": 64, "rms_norm_eps": 1e-5,
        "rope_theta": 500000.0, "vocab_size": 128256, "tie_word_embeddings": True, "initializer_range": 0.02,
        "rope_scaling": {"factor": 32.0, "high_freq_factor": 4.0, "low_freq_factor": 1.0,
                         "original_max_position_embeddings": 8192, "rope_type": "llama3"}})


def kimi_audio_7b_config(use_whisper_feature: bool = False):
    from touchnet_amd.models.kimi_audio import KimiAudioConfig
    # examples/audio/sft/asr/wenetspeech/config/Kimi-Audio-7B.json (the decoder keys; with `use_whisper_feature` also the
    # whisper-large-v3 speech encoder + VQ adaptor: KimiAudioConfig's defaults)
    return KimiAudioConfig.from_dict({
        "use_whisper_feature": use_whisper_feature,
        "hidden_size": 3584, "intermediate_size": 18944, "num_attention_heads": 28, "num_key_value_heads": 4,
        "num_hidden_layers": 28, "head_dim": 128, "rms_norm_eps": 1e-6, "rope_theta": 1000000.0, "vocab_size": 168448,
        "initializer_range": 0.02, "tie_word_embeddings": False, "kimia_mimo_layers": 6,
        "kimia_mimo_transformer_from_layer_index": 21, "kimia_token_offset": 152064})


# BASELINE.json configs: C = Qwen2-Audio-7B FSDP2 dp 8; D = long audio, CP 4 x dp 2; E = Kimi-Audio-7B, TP 2 x dp 4
RECIPES = {"qwen2_audio_7b_long": {"cp": 4}, "kimi_audio_7b": {"tp": 2}, "kimi_audio_7b_speech": {"tp": 2}}


def recipe_degrees(workload: str, gpus: int, cp, tp, emulate_rank=None):
    """`python bench.py --gpus N --workload W` without --cp / --tp runs W's BASELINE recipe: the degree the config names
    when N is a multiple of it (8 GPUs: long audio = cp 4 x dp 2, Kimi = tp 2 x dp 4, everything else dp 8), 1 otherwise
    (and always 1 on one GPU).  An explicit flag always wins."""
    rec = RECIPES.get(workload, {})
    pick = lambda given, key: given if given is not None else (
        rec[key] if (key in rec and emulate_rank is None and gpus > 1 and gpus % rec[key] == 0) else 1)
    return pick(cp, "cp"), pick(tp, "tp")


def parallel_layout(world: int, cp: int = 1, tp: int = 1, emulate_rank=None) -> dict:
    """How `--gpus / --cp / --tp / --emulate-rank` split the job: dp x cp x tp real ranks, or ONE process playing rank
    `emulate_rank` of a cp- or tp-way group."""
    if cp < 1 or tp < 1:
        raise SystemExit("--cp / --tp must be >= 1")
    if emulate_rank is not None:
        if world != 1:
            raise SystemExit("--emulate-rank plays one rank of the group on ONE GPU: use --gpus 1")
        if (cp > 1) == (tp > 1):
            raise SystemExit("--emulate-rank needs exactly one of --cp N / --tp N")
        n = cp * tp
        if not 0 <= emulate_rank < n:
            raise SystemExit(f"--emulate-rank {emulate_rank} outside the {n}-way group")
        kind = "cp" if cp > 1 else "tp"
        return {"dp": 1, "cp": cp, "tp": tp, "emulated": {"group": kind, "size": n, "rank": emulate_rank},
                "label": f"{kind}{n} (rank {emulate_rank} emulated on one GPU, exchanges skipped)"}
    if world % (cp * tp):
        raise SystemExit(f"--gpus {world} is not a multiple of cp x tp = {cp * tp}")
    dp = world // (cp * tp)
    parts = [f"cp{cp}"] if cp > 1 else []
    parts += [f"tp{tp}"] if tp > 1 else []
    parts += [f"dp{dp}" + (f" (gradients reduced / optimizer state sharded over dp x cp = {dp * cp})" if cp > 1 else "")] if dp * cp > 1 else []
    return {"dp": dp, "cp": cp, "tp": tp, "emulated": None, "label": " x ".join(parts) or "single-gpu"}


def tiny_text_config():
    from touchnet_amd.models.llama import DecoderConfig
    return DecoderConfig.from_dict({
        "model_type": "llama", "hidden_size": 256, "intermediate_size": 512, "num_attention_heads": 4,
        "num_hidden_layers": 2, "num_key_value_heads": 2, "head_dim": 64, "rms_norm_eps": 1e-5,
        "rope_theta": 500000.0, "vocab_size": 1024, "tie_word_embeddings": True, "initializer_range": 0.02})


class Workload:
    """Holds the device-resident inputs of one rank and produces the batch dict inside the timed step."""

    def __init__(self, name, device, rank, B=None, T=None, cp=None):
        """`rank` seeds the data (the DATA-parallel rank: cp / tp peers hold the same batch); `cp` = (cp, cp_rank): the
        long-audio workload hands every cp rank the 30 s clips that touch ITS part of the sequence — what a loader under
        context parallelism does once per batch, ahead of the step."""
        import touchnet_amd.functional as F
        from touchnet_amd.bin.train import TrainConfig
        from touchnet_amd.data import synthetic
        from touchnet_amd.models.touch_audio import TouchAudioConfig
        self.name, self.device, self.F = name, device, F
        seed = 2025 + rank
        self.job = TrainConfig()
        if name in ("qwen2_audio_7b", "qwen2_audio_7b_short"):
            short = name.endswith("_short")
            self.B, self.T = B or 2, T or 8192
            self.job.training_model_name = "qwen2_audio_mi355"
            self.job.lr_scheduler_lr = 2e-5
            self.model_config = qwen2_audio_7b_config()
            self.seq_cfg = self.model_config.text_config
            tok, n_audio = synthetic.qwen2_audio_plan(self.seq_cfg.vocab_size, self.model_config.audio_token_index,
                                                      self.B, self.T, seed, audio_seconds=(2.0, 14.5) if short else None)
            g = torch.Generator().manual_seed(seed)
            # 30 s-padded utterances (WhisperFeatureExtractor padding="max_length"): speech-shaped noise for
            # U[2, 14.5] s, zeros to 30 s.  The frontend runs on all 480 000 samples, like the reference.
            # qwen2_audio_7b (the metric's workload): every clip fills its 750 AUDIO tokens;  qwen2_audio_7b_short: the
            # token count follows the valid length like in the reference's processor (WenetSpeech-length utterances).
            wav = torch.zeros(n_audio, 480000)
            for i in range(n_audio):
                n = (tok["audio_samples"][i] if short
                     else int(float(torch.empty(1).uniform_(2.0, 14.5, generator=g)) * 16000))
                wav[i, :n] = (torch.randn(n, generator=g) * 0.1).clamp_(-1, 1)
            self.wav = wav.to(device)
            self.tokens = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in tok.items()
                           if k != "audio_samples"}
            n_tok = int(tok["audio_positions"].numel())
            self.data_desc = (f"synthetic: {n_audio} x 30 s-padded 16 kHz clips/GPU -> "
                              + (f"{n_tok} audio tokens in all (U[2, 14.5] s utterances, token count from the valid length)"
                                 if short else "750 audio tokens each")
                              + f", ~14-token prompt, U{{5..40}}-token transcripts, packed B={self.B} x T={self.T}")
        elif name == "qwen2_audio_7b_long":
            self.B, self.T = B or 1, T or 65536
            self.job.training_model_name = "qwen2_audio_mi355"
            self.job.lr_scheduler_lr = 2e-5
            self.model_config = qwen2_audio_7b_config()
            self.seq_cfg = self.model_config.text_config
            tok, n_audio = synthetic.qwen2_audio_long_plan(self.seq_cfg.vocab_size, self.model_config.audio_token_index,
                                                           self.B, self.T, seed)
            self.host_tokens = tok
            self.n_clips_global = n_audio
            clips = np.arange(n_audio)
            if cp is not None and cp[0] > 1:
                from touchnet_amd.utils.context_parallel import ContextParallel
                view = ContextParallel(None, self.T, emulate=(cp[0], cp[1]))
                clips, pos, rows = view.shard_audio(tok["audio_positions"].numpy(), tok["audio_output_lengths"].numpy(), 750)
                lab = tok["labels"]
                tok = dict(tok, audio_positions=torch.from_numpy(pos), audio_rows=torch.from_numpy(rows),
                           audio_output_lengths=tok["audio_output_lengths"][clips], audio_cp_sharded=True,
                           labelled_rows_max_cp=[int((ContextParallel(None, self.T, emulate=(cp[0], r)).shard(lab, 1)
                                                      != -100).sum()) for r in range(cp[0])])
            g = torch.Generator().manual_seed(seed)
            wav = (torch.randn(n_audio, 480000, generator=g) * 0.1).clamp_(-1, 1)    # consecutive 30 s windows: all speech
            self.wav = wav[torch.from_numpy(clips)].to(device)
            self.tokens = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in tok.items()}
            self.data_desc = (f"synthetic long audio: documents of up to 40 consecutive 30 s windows (20 min, 750 audio "
                              f"tokens each) + ~50 transcript tokens per window, packed B={self.B} x T={self.T}: "
                              f"{n_audio} clips per row group, {len(clips)} of them on this rank")
        elif name in ("kimi_audio_7b", "kimi_audio_7b_speech"):
            speech = name.endswith("_speech")
            self.B, self.T = B or (1 if speech else 2), T or 8192
            self.job.training_model_name = "kimi_audio_mi355"
            self.job.lr_scheduler_lr = 2e-5
            self.model_config = kimi_audio_7b_config(use_whisper_feature=speech)
            self.seq_cfg = self.model_config
            c = self.model_config
            tok = synthetic.kimi_audio_plan(c.kimia_token_offset, c.kimia_token_offset, c.vocab_size - c.kimia_token_offset,
                                            self.B, self.T, seed,
                                            media_markers=(c.kimia_media_begin, c.kimia_media_end) if speech else None)
            if speech:
                # the reference batch's `whisper_input_features`: one 30 s-padded clip per media-marker pair; the log-mel
                # runs on the device inside the step, like for the Qwen2-Audio workload
                g = torch.Generator().manual_seed(seed)
                clip_tokens = tok.pop("clip_tokens")
                n_clips = len(clip_tokens)
                wav = torch.zeros(n_clips, 480000)
                for i, na in enumerate(clip_tokens):
                    n = min(480000, max(int(na), 1) * 1280)
                    wav[i, :n] = (torch.randn(n, generator=g) * 0.1).clamp_(-1, 1)
                self.wav = wav.to(device)
            self.tokens = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in tok.items()}
            self.data_desc = ((f"synthetic Kimi-Audio SFT form: {n_clips} x 30 s-padded 16 kHz clips -> Whisper speech encoder "
                               f"(1500 frames each) + VQ adaptor, written between the media markers; " if speech else "")
                              + f"synthetic interleaved audio/text: U[2, 14.5] s of 12.5 Hz audio codes on the audio stream, "
                              f"then a U{{5..40}}-token transcript on the text stream (labels on the transcript, text head), "
                              f"packed B={self.B} x T={self.T}")
        elif name in ("llama_asr_1b", "tiny"):
            text = llama_1b_text_config() if name == "llama_asr_1b" else tiny_text_config()
            self.B, self.T = B or 1, T or (8192 if name == "llama_asr_1b" else 512)
            self.job.training_model_name = "touch_audio_mi355"
            self.model_config = TouchAudioConfig(text_config=text, input_size=400)
            self.seq_cfg = text
            self.utts, _ = synthetic.asr_waveforms(self.T, self.B, seed)
            g = torch.Generator().manual_seed(seed)
            self.wavs = [(torch.randn(n, generator=g) * 0.1).clamp_(-1, 1).to(device) for n, _ in self.utts]
            self.seed = seed
            self.data_desc = (f"synthetic AISHELL-shaped: U[1.5,14.5] s 16 kHz clips, U{{4..25}}-token "
                              f"transcripts, fbank80 stack5/stride4, packed B={self.B} x T={self.T}")
        else:
            raise SystemExit(f"unknown workload {name}")

    def make_batch(self):
        """Runs the device frontend and returns the batch dict (everything already on the device)."""
        F = self.F
        if self.name == "kimi_audio_7b":
            return dict(self.tokens)                 # (discrete codes: the frozen VQ tokenizer is the loader's, out of scope)
        if self.name == "kimi_audio_7b_speech":
            mel = torch.stack([F.log_mel_spectrogram(w, 128) for w in self.wav])       # [n, 3000, 128]
            batch = dict(self.tokens)
            batch["whisper_input_features"] = mel.transpose(1, 2)
            return batch
        if self.name.startswith("qwen2_audio_7b"):
            mel = torch.stack([F.log_mel_spectrogram(w, 128) for w in self.wav])       # [n, 3000, 128]
            batch = dict(self.tokens)
            batch["input_features"] = mel.transpose(1, 2)                              # [n, 128, 3000]
            return batch
        from touchnet_amd.data.synthetic import asr_batch_from_device_frontend
        batch, _, _ = asr_batch_from_device_frontend(self.seq_cfg.vocab_size, self.B, self.T, self.device,
                                                     seed=self.seed, frontend=F, wavs=self.wavs, utts=self.utts)
        return batch


def cpu_baseline(workload: "Workload", seconds_budget: float = 20.0, n_params: int = 0):
    """The reference's CPU path timed beside the GPU number: it has no CPU training entry point
    (SURVEY.md §0 fact 5), so this is the ORACLE restatement (oracle/nn.py + oracle/loss.py: eager HF maths +
    reference loss) in fp32 on the host cores — a bounded sample: ONE decoder block forward+backward at the
    workload's widths (T = 1024 and 2048) plus the lm_head+CE on 64 tokens, extrapolated x layers to tokens/s; for the
    Qwen2-Audio workloads also one audio-tower layer on one clip and torch's AdamW on a parameter sample (x layers x clips,
    x parameters), so that the figure estimates the whole step of config C.
    It is a reported baseline ("port"), never the measured product path."""
    from oracle import loss as oloss
    from oracle import nn as onn
    cfg = workload.seq_cfg
    cores = min(os.cpu_count() or 1, 64)          # eager fp32 at these widths stops scaling past ~64 threads
    torch.set_num_threads(cores)
    H, I, Nh, Nkv, D, V = (cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads,
                           cfg.num_key_value_heads, cfg.head_dim, cfg.vocab_size)
    g = torch.Generator().manual_seed(0)
    sd = {"l.input_layernorm.weight": torch.ones(H), "l.post_attention_layernorm.weight": torch.ones(H)}
    for n, (o, i) in {"self_attn.q_proj": (Nh * D, H), "self_attn.k_proj": (Nkv * D, H),
                      "self_attn.v_proj": (Nkv * D, H), "self_attn.o_proj": (H, Nh * D),
                      "mlp.gate_proj": (I, H), "mlp.up_proj": (I, H), "mlp.down_proj": (H, I)}.items():
        sd[f"l.{n}.weight"] = (torch.randn(o, i, generator=g) * 0.02).requires_grad_()
    c = dict(num_attention_heads=Nh, num_key_value_heads=Nkv, head_dim=D, rms_norm_eps=cfg.rms_norm_eps)
    inv = onn.rope_inv_freq(D, cfg.rope_theta, cfg.rope_scaling)
    def time_block(Tn, budget):
        doc = torch.ones(1, Tn, dtype=torch.int64)
        cos, sin = onn.rope_cos_sin(torch.arange(Tn)[None], inv, torch.float32)
        allow = onn.doc_causal_allow(doc)
        xx = torch.randn(1, Tn, H, generator=g).requires_grad_()

        def block():
            onn.decoder_layer(sd, "l.", c, xx, cos, sin, allow).sum().backward()
        block()                                                        # (warm-up: allocator, thread pool)
        ts = []
        for _ in range(budget):                                        # a FIXED number of repeats, not a time budget
            t0 = time.perf_counter()
            block()
            ts.append((time.perf_counter() - t0) / Tn)                 # s per token per layer
        return sorted(ts)

    # eager attention materialises the T x T scores whatever the documents are (the reference's CPU-capable path has no
    # block sparsity), so its cost per token is a + b*T: two lengths give a and b, evaluated at the workload's T
    # Pinned method: min(cores, 64) threads, T = 1024 and 2048, 5 timed repeats each behind one warm-up,
    # MEDIANS extrapolated; the spread of the repeats is reported beside the value (the number moved 6.4 .. 13.8 tok/s
    # between rounds with the box and the time-budgeted repeat count: it is a baseline, never a ratio to quote).
    T1, T2, REP = 1024, 2048, 5
    r1, r2 = time_block(T1, REP), time_block(T2, REP)
    t1, t2 = r1[REP // 2], r2[REP // 2]
    slope = max(0.0, (t2 - t1) / (T2 - T1))
    t_block = t1 + slope * (workload.T - T1)
    spread = max(r1[-1] / r1[0], r2[-1] / r2[0]) - 1.0
    Th = 64
    w = (torch.randn(V, H, generator=g) * 0.02).requires_grad_()
    hh = torch.randn(1, Th, H, generator=g).requires_grad_()
    labels = torch.randint(0, V, (1, Th), generator=g)

    def head():
        ps, _ = oloss.cross_entropy_loss(torch.nn.functional.linear(hh, w), labels, torch.full((1, Th), 8), 8)
        ps.backward()
    head()
    hs = []
    for _ in range(REP):
        t0 = time.perf_counter()
        head()
        hs.append((time.perf_counter() - t0) / Th)
    t_head = sorted(hs)[REP // 2]
    per_token = t_block * cfg.num_hidden_layers + t_head
    extra = "audio tower and optimizer excluded -> an upper bound on CPU throughput"
    ac = getattr(workload.model_config, "audio_config", None)
    if (workload.name.startswith("qwen2_audio_7b") and ac is not None and getattr(workload, "wav", None) is not None
            and n_params > 0):
        # (round 6) the rest of the step, so that the figure is an estimate of config C and not of its
        # decoder: ONE Whisper encoder layer forward + backward on one clip's 1500 frames (oracle.nn.whisper_layer: the
        # reference's patched tower, touchnet/models/qwen2_audio/__init__.py:18-133) x layers x clips, and torch's AdamW
        # arithmetic (fp32, foreach form) on a 32 M-parameter sample x parameters / sample
        C, heads, ffn, frames = ac.d_model, ac.encoder_attention_heads, ac.encoder_ffn_dim, 1500
        tsd = {}
        for n, (o, i) in {"self_attn.q_proj": (C, C), "self_attn.k_proj": (C, C), "self_attn.v_proj": (C, C),
                          "self_attn.out_proj": (C, C), "fc1": (ffn, C), "fc2": (C, ffn)}.items():
            tsd[f"t.{n}.weight"] = (torch.randn(o, i, generator=g) * 0.02).requires_grad_()
            if n != "self_attn.k_proj":
                tsd[f"t.{n}.bias"] = torch.zeros(o, requires_grad=True)
        for n in ("self_attn_layer_norm", "final_layer_norm"):
            tsd[f"t.{n}.weight"], tsd[f"t.{n}.bias"] = torch.ones(C, requires_grad=True), torch.zeros(C, requires_grad=True)
        hx = torch.randn(1, frames, C, generator=g).requires_grad_()
        allow = torch.ones(1, frames, frames, dtype=torch.bool).tril_()

        def tower_layer():
            onn.whisper_layer(tsd, "t.", hx, heads, allow).sum().backward()
        tower_layer()
        tt = []
        for _ in range(REP):
            t0 = time.perf_counter()
            tower_layer()
            tt.append(time.perf_counter() - t0)
        t_tower = sorted(tt)[REP // 2] * ac.encoder_layers * int(workload.wav.shape[0])        # s per step
        n_s = 32 << 20
        pw = torch.zeros(n_s, requires_grad=True)
        pw.grad = torch.full((n_s,), 1e-3)
        opt = torch.optim.AdamW([pw], lr=1e-4, foreach=True)
        opt.step()
        to = []
        for _ in range(REP):
            t0 = time.perf_counter()
            opt.step()
            to.append(time.perf_counter() - t0)
        t_opt = sorted(to)[REP // 2] * n_params / n_s                                            # s per step
        tokens = workload.B * workload.T
        per_token += (t_tower + t_opt) / tokens
        extra = (f"+ 1 Whisper encoder layer fwd+bwd on 1500 frames x{ac.encoder_layers} layers x{int(workload.wav.shape[0])} clips "
                 f"({t_tower:.1f} s/step) + torch AdamW on a {n_s >> 20} M-parameter sample x {n_params / 1e9:.2f} G parameters "
                 f"({t_opt:.1f} s/step); conv stem, projector, log-mel and gradient clipping excluded")
    return {"value": round(1.0 / per_token, 2), "unit": "tokens/s (extrapolated)", "cores": cores, "kind": "port",
            "repeats": REP, "repeat_spread": round(spread, 3),
            "sample": f"oracle fp32 eager, {cores} threads: 1 decoder block fwd+bwd at T={T1} and T={T2}, median of {REP} repeats "
                      f"({t1 * 1e3:.3f} / {t2 * 1e3:.3f} ms per token per layer), per-token cost a + b*T evaluated at the "
                      f"workload's T={workload.T}, x{cfg.num_hidden_layers} layers, + lm_head/CE on {Th} tokens; " + extra}


def executed_flops_per_gpu(wl: "Workload", trainer, layout: dict, lm_head_rows: int) -> float:
    """FLOPs ONE GPU of a cp / tp layout executes per step (the sharded workloads' counterpart of the headline's
    `step_mfu_executed_flops`): 6 x parameters x rows for every GEMM on the rows it actually runs on, attention on the
    allowed (query, key) pairs of this rank's query rows (position_ids + 1 keys per query), fwd + 2.5x bwd."""
    c, cp, tp = wl.seq_cfg, layout["cp"], layout["tp"]
    tok = getattr(wl, "host_tokens", None) or {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in wl.tokens.items()}
    pos, doc = tok["position_ids"], tok["attention_mask"]
    if cp > 1:
        from touchnet_amd.utils.context_parallel import ContextParallel
        view = ContextParallel(None, wl.T, emulate=(cp, trainer.cp.rank if trainer.cp is not None else 0))
        pos, doc = view.shard(pos, 1), view.shard(doc, 1)
    rows = pos.numel()
    pairs = int(((pos + 1) * (doc > 0)).sum())
    H, I, Nh, Nkv, D, L = (c.hidden_size, c.intermediate_size, c.num_attention_heads, c.num_key_value_heads, c.head_dim,
                           c.num_hidden_layers)
    layer = H * Nh * D * 2 + 2 * H * Nkv * D + 3 * H * I
    fl = 6.0 * layer * L * rows / tp
    fl += 3.5 * 4.0 * D * (Nh / tp) * pairs * L
    fl += 6.0 * c.vocab_size * H * lm_head_rows / (tp if wl.job.training_enable_loss_parallel else 1)
    if hasattr(wl, "wav") and wl.name == "kimi_audio_7b_speech":
        # the speech encoder runs on all 1500 frames of every 30 s-padded clip (like the reference: transformers'
        # WhisperEncoder ignores the frame mask), BIDIRECTIONAL attention: every (query, key) pair of a clip
        ac = wl.model_config.speech_encoder_dims()
        enc = sum(p.numel() for p in trainer.model.speech_encoder.parameters())
        n = wl.wav.shape[0]
        fl += 6.0 * enc * n * 1500 + 3.5 * 4.0 * 64 * ac.encoder_attention_heads * n * (1500 * 1500) * ac.encoder_layers
        fl += 6.0 * (4 * ac.d_model * H + H * H) * n * 375                      # VQ adaptor on the x4-stacked frames
    elif hasattr(wl, "wav"):
        ac = wl.model_config.audio_config
        tower = sum(p.numel() for p in trainer.model.audio_tower.parameters())
        n = wl.wav.shape[0]
        fl += 6.0 * tower * n * 1500 + 3.5 * 4.0 * 64 * ac.encoder_attention_heads * n * (1500 * 1501 // 2) * ac.encoder_layers
        fl += 6.0 * ac.d_model * H * n * 750
    return fl


def decoder_rows(workload: "Workload") -> int:
    """Rows the decoder's GEMMs and row kernels run on: the non-pad slots of the packed batch (the packers'
    `valid_rows_max`, rounded up to the GEMM tile height) when the model drops the padding slots
    (models/llama/modeling_llama.py DecoderModel._drop_pad_rows), else all B x T."""
    import touchnet_amd.models.llama.modeling_llama as _ml
    full = workload.B * workload.T
    v = getattr(workload, "tokens", {}).get("valid_rows_max") if hasattr(workload, "tokens") else None
    if v is None or not _ml.SKIP_PAD_ROWS or not workload.job.training_enable_fused_ce:
        return full
    mc = min((int(v) + 255) // 256 * 256, full)
    return mc if mc + 256 <= full else full


def allowed_attention_pairs(workload: "Workload") -> int:
    """(query, key) pairs the document-causal mask allows in one batch of the workload: sum over documents of n (n + 1) / 2"""
    doc = (workload.tokens["attention_mask"] if hasattr(workload, "tokens")
           else torch.ones(workload.B, workload.T, dtype=torch.int64))
    allowed = 0
    for row in doc.cpu().numpy():
        _, counts = np.unique(row[row > 0], return_counts=True)
        allowed += int(sum(int(c) * (int(c) + 1) // 2 for c in counts))
    return allowed


def kernel_rooflines(workload: "Workload"):
    """Live HIP-event timings (on torch's current stream = the stream the C ABI is given) of the hand-written
    kernels at this workload's shapes: achieved algorithmic bytes/flops per launch vs the roofline."""
    F, dev, cfg = workload.F, workload.device, workload.seq_cfg
    B, T = workload.B, workload.T
    N, H, I, Nh, Nkv, D = (decoder_rows(workload), cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads,
                           cfg.num_key_value_heads, cfg.head_dim)
    bf = torch.bfloat16
    out = []

    def t_ms(fn, it=10):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / it

    L_ = cfg.num_hidden_layers

    def add(name, ms, bytes_=None, flops=None, per_step=0):
        """per_step = launches of this kernel (at this shape) in one training step of the workload (0 = comparison only)"""
        if bytes_ is not None:
            a = bytes_ / (ms * 1e-3) / 1e9
            out.append({"kernel": name, "bound": "hbm", "ms": round(ms, 4), "achieved": round(a, 1),
                        "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(a * 1e9 / HBM_PEAK, 3),
                        "launches_per_step": per_step})
        else:
            a = flops / (ms * 1e-3) / 1e12
            out.append({"kernel": name, "bound": "mfma", "ms": round(ms, 4), "achieved": round(a, 1),
                        "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": round(a * 1e12 / MFMA_PEAK, 3),
                        "launches_per_step": per_step})
    x = torch.randn(N, H, dtype=bf, device=dev)
    r = torch.randn(N, H, dtype=bf, device=dev)
    w = torch.ones(H, dtype=bf, device=dev)
    add("add+rmsnorm fwd", t_ms(lambda: F.rms_norm(x, w, 1e-5, residual=r)), bytes_=4 * N * H * 2, per_step=2 * L_)
    g_, u_ = torch.randn(N, I, dtype=bf, device=dev), torch.randn(N, I, dtype=bf, device=dev)
    fused_mlp = F.LINEAR_GEMM == "own" and F.MLP_EPILOGUE
    add("swiglu fwd (standalone kernel; in the step it is the epilogue of the gate+up launch)" if fused_mlp else "swiglu fwd",
        t_ms(lambda: F.swiglu(g_, u_)), bytes_=3 * N * I * 2, per_step=0 if fused_mlp else L_)
    del g_, u_
    doc = workload.tokens["attention_mask"] if hasattr(workload, "tokens") else torch.ones(B, T, device=dev)
    mask = F.build_packed_mask(doc)
    allowed = allowed_attention_pairs(workload)
    q = torch.randn(B, T, Nh, D, dtype=bf, device=dev)
    k = torch.randn(B, T, Nkv, D, dtype=bf, device=dev)
    v = torch.randn(B, T, Nkv, D, dtype=bf, device=dev)
    fl = 4.0 * D * Nh * allowed
    add("packed attention fwd (true masked flops)", t_ms(lambda: F.packed_attention(q, k, v, mask)), flops=fl, per_step=L_)
    qg, kg, vg = [t.clone().requires_grad_() for t in (q, k, v)]
    o = F.packed_attention(qg, kg, vg, mask)
    do = torch.randn_like(o)
    add("packed attention bwd (true masked flops)",
        t_ms(lambda: torch.autograd.grad(o, [qg, kg, vg], do, retain_graph=True)), flops=2.5 * fl, per_step=L_)
    del q, k, v, qg, kg, vg, o, do
    # the step's dominant kernels are the linear layers' GEMMs (~70 % of its time): one of each kind at the MLP shapes, on
    # the path that is configured (hand-written kernel in its native operand modes, or the library on transposed copies),
    # the other path beside it for comparison (launches_per_step = 0)
    own = F.LINEAR_GEMM == "own"
    wg = torch.randn(I, H, dtype=bf, device=dev)
    wu = torch.randn(I, H, dtype=bf, device=dev)
    dy = torch.randn(N, I, dtype=bf, device=dev)
    du = torch.randn(N, I, dtype=bf, device=dev)
    fl1 = 2.0 * N * H * I
    fz = own and fused_mlp
    add(f"tn::gemm fwd gate_proj [{N}x{H}]x[{I}x{H}]^T (hand-written, csrc/gemm.hip)", t_ms(lambda: F.gemm([(x, wg)])),
        flops=fl1, per_step=(L_ if fz else 3 * L_) if own else 0)     # down_proj (same flops); unfused: gate, up, down
    add("tn::gemm dgrad gate+up: dX = dG Wg + dU Wu, ONE two-segment launch, W read contraction-major",
        t_ms(lambda: F.gemm([(dy, wg), (du, wu)], b_kmaj=True)), flops=2 * fl1, per_step=L_ if own else 0)
    grouped = own and F.GROUPED_WGRAD
    add("tn::gemm wgrad gate_proj: dW = dG^T X, both operands contraction-major (no transposed copies)",
        t_ms(lambda: F.gemm([(dy, x)], True, True)), flops=fl1, per_step=(0 if grouped else 3 * L_) if own else 0)
    if own:
        # the round-5 launches of the MLP (csrc/gemm.hip EPI_SWIGLU_FWD / EPI_SWIGLU_BWD / EPI_GROUPED)
        add("tn::gemm fwd gate+up with the SwiGLU epilogue: gate, up, act = silu(gate)*up from ONE launch",
            t_ms(lambda: F.gemm_swiglu_fwd(x, wg, wu)), flops=2 * fl1, per_step=L_ if fz else 0)
        wd = torch.randn(H, I, dtype=bf, device=dev)
        dyh = torch.randn(N, H, dtype=bf, device=dev)
        add("tn::gemm d(act) = dY Wd with the SwiGLU-backward epilogue: d(gate), d(up) from ONE launch, d(act) never in HBM",
            t_ms(lambda: F.gemm_swiglu_bwd(dyh, wd, dy, du)), flops=fl1, per_step=L_ if fz else 0)
        add("tn::gemm wgrad gate+up+down GROUPED: three weight gradients as ONE launch, remainder of the tile list split-K",
            t_ms(lambda: F.gemm_grouped_wgrad([(dy, x), (du, x), (dyh, dy)])), flops=3 * fl1, per_step=L_ if grouped else 0)
        del wd, dyh
    add(f"hipBLASLt GEMM fwd gate_proj, same shape", t_ms(lambda: torch.nn.functional.linear(x, wg)),
        flops=fl1, per_step=0 if own else 3 * L_)
    wgt = F.transpose_2d(wg)
    add("hipBLASLt GEMM dgrad gate_proj (W pre-transposed)", t_ms(lambda: torch.mm(dy, wgt.t())), flops=fl1,
        per_step=0 if own else 3 * L_)
    dyt = torch.empty(2 * I, N, dtype=bf, device=dev)
    F.transpose_2d(dy, out=dyt[:I])
    F.transpose_2d(dy, out=dyt[I:])
    xt = F.transpose_2d(x)
    add("hipBLASLt GEMM wgrad gate+up fused (both operands pre-transposed)", t_ms(lambda: torch.mm(dyt, xt.t())),
        flops=2 * fl1, per_step=0 if own else L_)
    add("autograd-layout wgrad gate_proj (dY^T X, for comparison)", t_ms(lambda: torch.mm(dy.t(), x)), flops=fl1)
    add("bf16 transpose (tn_transpose_bf16) [N, I] (library path only)", t_ms(lambda: F.transpose_2d(dy, out=dyt[:I])),
        bytes_=4 * N * I)
    return out, allowed


def self_launch(gpus: int, argv=None) -> None:
    """`python bench.py --gpus N` (N > 1) WITHOUT a launcher: re-execute this command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
    (one rank per GPU over RCCL) and exit with its status.  A no-op when a launcher already set WORLD_SIZE or N == 1.
    Fewer GPUs than ranks: refused, unless TN_DIST_BACKEND=gloo (ranks share GPUs, device buffers staged through the host:
    the one-GPU development path, see touchnet_amd/utils/distributed.py)."""
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < gpus and os.environ.get("TN_DIST_BACKEND") != "gloo":
        raise SystemExit(f"bench.py --gpus {gpus}: {have} GPU(s) visible (RCCL needs one device per rank; "
                         f"TN_DIST_BACKEND=gloo lets ranks share a GPU for functional runs)")
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__),
           *(sys.argv[1:] if argv is None else argv)]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("TN_BENCH_WORKLOAD", "qwen2_audio_7b"))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--seqlen", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-rooflines", action="store_true")
    ap.add_argument("--unfused-ce", action="store_true")
    ap.add_argument("--ce-chunk", type=int, default=None, help="tokens per lm_head+CE chunk (default: TrainConfig's)")
    ap.add_argument("--compact-lm-head", action="store_true",
                    help="lm_head + CE on labelled positions with the EXACT count read back (one host sync per step)")
    ap.add_argument("--all-rows-lm-head", action="store_true",
                    help="A/B switch: ignore the loader's labelled-row bound and run lm_head + CE on all B*T positions")
    ap.add_argument("--no-gemm-tuning", action="store_true", help="library-default GEMM algorithm selection")
    ap.add_argument("--linear-gemm", choices=("lib", "own"), default=None,
                    help="own (default) = the linear layers' GEMMs on the hand-written MFMA kernel (csrc/gemm.hip) in its "
                         "native operand modes; lib = hipBLASLt on transposed copies (A/B runs; TN_LINEAR_GEMM)")
    ap.add_argument("--cp", type=int, default=None,
                    help="context-parallel degree (ranks split as dp x cp x tp); default: the workload's BASELINE recipe "
                         "when --gpus allows it (qwen2_audio_7b_long: cp 4), else 1")
    ap.add_argument("--tp", type=int, default=None,
                    help="tensor-parallel degree; default: the workload's BASELINE recipe (kimi_audio_7b*: tp 2), else 1")
    ap.add_argument("--emulate-rank", type=int, default=None,
                    help="with --gpus 1 and --cp N or --tp N: run rank r of the N-way group alone on one GPU")
    ap.add_argument("--ac", choices=("none", "full", "selective", "op"), default="none",
                    help="activation checkpointing: full = every block, selective = every 2nd block, op = the reference's "
                         "op-level policy (keep GEMM / attention outputs, recompute the row kernels)")
    ap.add_argument("--wgrad-stream", action="store_true",
                    help="weight-gradient GEMMs on a side stream beside the input-gradient chain, one workgroup per tile "
                         "(TN_WGRAD_STREAM=1): A/B switch, see functional.enable_wgrad_stream")
    ap.add_argument("--loss-parallel", action="store_true",
                    help="with --tp: vocabulary-parallel lm_head + CE (the reference's enable_loss_parallel)")
    ap.add_argument("--no-sequence-parallel", action="store_true",
                    help="with --tp: keep the residual stream / norms replicated (all-reduce per block output)")
    ap.add_argument("--emulate-shards", type=int, default=0,
                    help="with --emulate-rank: shard the optimizer state / gradient buckets like rank 0 of a data-parallel "
                         "group of this size (flat engine, collectives replaced by local copies): a real rank's MEMORY")
    ap.add_argument("--dp-engine", choices=("flat", "fsdp2"), default=None,
                    help="data parallelism for N > 1: flat (default) = utils/zero_dp.py, flat per-block buffers + sharded "
                         "optimizer state; fsdp2 = torch fully_shard as the reference applies it (TN_DP_ENGINE)")
    args = ap.parse_args()
    args.cp, args.tp = recipe_degrees(args.workload, args.gpus, args.cp, args.tp, args.emulate_rank)
    layout = parallel_layout(args.gpus, args.cp, args.tp, args.emulate_rank)
    self_launch(args.gpus)
    if args.linear_gemm:
        import touchnet_amd.functional as _F
        _F.LINEAR_GEMM = args.linear_gemm

    if args.wgrad_stream or os.environ.get("TN_WGRAD_STREAM") == "1":
        os.environ.setdefault("TN_GEMM_PERSIST", "0")
    from touchnet_amd.utils import gemm_tuning
    tuned = (not args.no_gemm_tuning) and gemm_tuning.enable()

    import touchnet_amd.specs  # noqa: F401  (registers the TrainSpecs)
    from touchnet_amd.bin.train import Trainer
    from touchnet_amd.utils.distributed import build_dp_mesh, init_distributed

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    rank, local, world = init_distributed("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device = torch.device("cuda", local)
    forced = os.environ.get("TN_FORCE_FSDP") == "1"     # (development: FSDP2 over a 1-rank RCCL mesh on one GPU)
    dp_mesh = cp_mesh = tp_mesh = fsdp_mesh = cp_emulate = None
    dp_rank, cp_view = rank, None
    emu = layout["emulated"]
    if emu:
        if args.emulate_shards > 1:
            from touchnet_amd.utils.zero_dp import EmulatedShardMesh
            fsdp_mesh = EmulatedShardMesh(args.emulate_shards, 0)
        if emu["group"] == "cp":
            cp_emulate = cp_view = (emu["size"], emu["rank"])
        else:
            from touchnet_amd.models.tensor_parallel import EmulatedTPMesh
            tp_mesh = EmulatedTPMesh(emu["size"], emu["rank"])
    elif args.cp > 1 or args.tp > 1:
        from touchnet_amd.utils.distributed import ParallelDims
        mesh = ParallelDims(dp_replicate=1, dp_shard=layout["dp"], cp=args.cp, tp=args.tp, pp=1,
                            world_size=world).build_mesh("cuda")
        names = mesh.mesh_dim_names
        dp_mesh = mesh["dp"] if "dp_shard" in names else None
        cp_mesh = mesh["cp"] if "cp" in names else None
        tp_mesh = mesh["tp"] if "tp" in names else None
        fsdp_mesh = mesh["dp_shard_cp"] if ("dp_shard" in names or "cp" in names) else None
        dp_rank = dp_mesh.get_local_rank() if dp_mesh is not None else 0
        if cp_mesh is not None:
            cp_view = (args.cp, cp_mesh.get_local_rank())
    else:
        dp_mesh = build_dp_mesh("cuda", world) if (world > 1 or forced) else None
    mesh = dp_mesh

    if args.wgrad_stream or os.environ.get("TN_WGRAD_STREAM") == "1":
        import touchnet_amd.functional as _F2
        _F2.enable_wgrad_stream()
    wl = Workload(args.workload, device, dp_rank, args.batch, args.seqlen, cp=cp_view)
    wl.job.training_enable_fused_ce = not args.unfused_ce
    wl.job.training_ce_compact_rows = args.compact_lm_head
    wl.job.training_activation_checkpoint_mode = "selective" if args.ac == "op" else args.ac
    if args.ac == "op":
        wl.job.training_activation_checkpoint_selective_ac_option = "op"
    if args.dp_engine:
        wl.job.training_dp_engine = args.dp_engine
    wl.job.training_enable_loss_parallel = bool(args.loss_parallel and args.tp > 1)
    wl.job.training_tp_sequence_parallel = not args.no_sequence_parallel
    if args.all_rows_lm_head and hasattr(wl, "tokens"):
        wl.tokens.pop("labelled_rows_max", None)
    if args.ce_chunk:
        wl.job.training_ce_chunk_tokens = args.ce_chunk
    trainer = Trainer(wl.job, wl.model_config, device, dp_mesh=dp_mesh, cp_mesh=cp_mesh, fsdp_mesh=fsdp_mesh,
                      tp_mesh=tp_mesh, cp_emulate=cp_emulate)

    def step():
        batch = trainer.next_batch(wl.make_batch())
        return trainer.train_step(batch)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        stats = step()
    fence()
    torch.cuda.reset_peak_memory_stats()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        stats = step()
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    ev_ms = ev0.elapsed_time(ev1) / args.steps

    # reference convention: labels.numel() per data-parallel rank (train.py:345); the cp / tp peers of a rank work on the
    # SAME rows, so the job's tokens are dp x B x T.  An emulated rank reports its own share of them (1/cp of the
    # sequence; 1/tp of the model on all tokens -> the tokens are credited 1/tp as well): per-GPU numbers throughout.
    share = layout["cp"] * layout["tp"]
    tokens_per_step = wl.B * wl.T * layout["dp"] / (share if emu else 1)
    tps = tokens_per_step * args.steps / elapsed
    gpus_in_job = 1 if emu else world
    if wl.name.startswith("kimi_audio_7b"):
        # the recipe trains the TEXT head: the mimo branch (6 layers + the audio head) is not executed, and the reference
        # formula (kimi_audio/__init__.py:63-80: L + L_mimo layers, all parameters) must not be credited with it
        fpt = trainer.spec.get_num_flop_per_token_fn(trainer.num_params_wo_emb, wl.model_config, wl.T, with_mimo=False)
    else:
        fpt = trainer.spec.get_num_flop_per_token_fn(trainer.num_params_wo_emb, wl.model_config, wl.T)
    mfu = fpt * (tps / gpus_in_job) / MFMA_PEAK
    nonpad = int((wl.tokens["attention_mask"] > 0).sum()) if hasattr(wl, "tokens") else None
    lrm = None if args.all_rows_lm_head else wl.make_batch().get("labelled_rows_max")     # what the packer told the model
    loss = float(stats["loss_per_sample"])
    loss_job = loss
    # what the job really ran on: read from the LIVE process group, with a one-element all-reduce over every rank (a
    # SCALE line can then be checked without trusting the command line: backend nccl == RCCL on ROCm, N ranks answered)
    dist_info = {"initialized": bool(dist.is_available() and dist.is_initialized())}
    if dist_info["initialized"]:
        ones = torch.ones(1, dtype=torch.float32, device=device)
        dist.all_reduce(ones)
        dist_info.update(backend=str(dist.get_backend()), world_size=int(dist.get_world_size()),
                         ranks_answering_all_reduce=int(round(float(ones))),
                         devices_visible=int(torch.cuda.device_count()))
        # rank 0's loss is ITS share (the per-sentence normalisation counts the sentences of all data- and context-parallel
        # ranks, touchnet/loss/cross_entropy.py:12-50 under train.py's all-reduced num_sentence); the job's loss — what one
        # GPU would print on the joined batch — is the sum over those ranks (tensor-parallel peers hold the same value)
        lsum = torch.tensor([loss], dtype=torch.float64, device=device).float()
        dist.all_reduce(lsum)
        loss_job = float(lsum) / (1 if emu else max(int(layout["tp"]), 1))
    try:
        v = torch.cuda.nccl.version()
        dist_info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:                                # (never lose the headline number to a diagnostics failure)
        dist_info["rccl_version"] = None
    if rank == 0:
        line = {
            "metric": "audio+text tokens/sec/node (packed seq, full train step) + step MFU",
            "value": round(tps, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": wl.data_desc + "; random-init weights",
            "config": {"workload": wl.name, "model": wl.job.training_model_name, "global_batch": wl.B * layout["dp"],
                       "seq_len": wl.T,
                       "parallelism": (layout["label"] if (args.cp > 1 or args.tp > 1) else
                                       f"fsdp2-dp{world}" if ((world > 1 or forced) and trainer.dp_engine is None) else
                                       f"dp{world}" if (world > 1 or forced) else "single-gpu"),
                       "dp_engine": ("flat per-block buffers: bf16 parameters replicated, one reduce-scatter (fp32) + one "
                                     "in-place all-gather per block, optimizer state sharded (utils/zero_dp.py)"
                                     if trainer.dp_engine is not None else
                                     "torch FSDP2 fully_shard per block" if (world > 1 or forced or args.cp > 1) else None),
                       **({"emulated": dict(emu, note="ONE rank of the group on one MI355X: its shards, kernels and share "
                                                      "of the data; exchanges with the peers skipped (cp: stand-in K/V "
                                                      "chunks of the same shape and document ids).  `value` is this GPU's "
                                                      "share of the tokens per second, not a group measurement")}
                          if emu else {}),
                       **({"tensor_parallel_plan": ("sequence parallel" if wl.job.training_tp_sequence_parallel else
                                                    "replicated residual stream")
                                                   + (" + vocabulary-parallel lm_head/CE" if wl.job.training_enable_loss_parallel
                                                      else ", lm_head replicated")} if args.tp > 1 else {}),
                       "activation_checkpointing": args.ac,
                       "params": trainer.num_params, "flop_per_token": fpt,
                       "fused_linear_ce": wl.job.training_enable_fused_ce,
                       "lm_head_rows": ("labelled only (exact count, host sync)" if args.compact_lm_head else
                                        f"labelled only: static bound {lrm} from the packer, no host sync"
                                        if (lrm is not None and wl.job.training_enable_fused_ce) else "all B*T positions"),
                       "last_layer_rows": ("labelled only behind the attention core (o_proj, MLP, norms)"
                                           if (lrm is not None and wl.job.training_enable_fused_ce
                                               and __import__("touchnet_amd.models.llama.modeling_llama", fromlist=["x"]).LAST_LAYER_LABELLED_ROWS)
                                           else "all B*T positions"),
                       "gemm_algorithms": "TunableOp replay (touchnet_amd/tuning)" if tuned else "library default",
                       "linear_layer_gemm": ("hand-written MFMA kernel (csrc/gemm.hip): forward, input-gradient and "
                                             "weight-gradient products in native operand modes, SwiGLU in the gate/up and "
                                             "down-dgrad epilogues, grouped MLP weight gradients, bias gradients from the "
                                             "weight-gradient launches; hipBLASLt only for outputs below 96 tiles (lm_head)"
                                             if __import__("touchnet_amd.functional", fromlist=["x"]).LINEAR_GEMM == "own"
                                             else "hipBLASLt (A/B mode)")},
            "step_mfu": round(mfu, 4),
            "mfu_convention": "6*N_wo_emb + 12*L*H*Dh*T per token (touchnet/models/*/__init__.py), no causal/packing "
                              "discount, no recompute credit, tokens = all B*T slots incl. pad (the reference's tps counts "
                              "them, train.py:345) — `roofline.frac` counts only the FLOPs that are executed",
            "nonpad_tokens_per_step_rank0": nonpad,
            # the decoder drops the padding slots (config.decoder_rows); `value` follows the reference's convention and
            # counts them: the same job in slots that carry a token (rank 0's count x data-parallel ranks), ADVICE r5
            "nonpad_tokens_per_s": (round(nonpad * layout["dp"] / (share if emu else 1) * args.steps / elapsed, 1)
                                    if nonpad is not None else None),
            "dist": dist_info,
            "loss_per_sample_last": round(loss, 5), "loss_per_sample_job": round(loss_job, 5), "hip_event_ms_per_step_rank0": round(ev_ms, 2),
            "peak_mem_GB_rank0": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
            **({"emulated_state_shards": args.emulate_shards} if args.emulate_shards > 1 else {}),
            "roofline": {"bound": "mfma", "achieved": round(fpt * (tps / gpus_in_job) / 1e12, 1), "peak": MFMA_PEAK / 1e12,
                         "unit": "TFLOP/s", "frac": round(mfu, 4), "traffic": None,
                         "note": "whole training step per GPU against the dense bf16 MFMA peak (reference MFU "
                                 "formula); per-kernel rooflines of the hand-written HIP kernels in `kernels`"},
        }
        if wl.name.startswith("kimi_audio_7b"):
            line["mfu_convention"] = ("6*N + 12*L*H*Dh*T per token over the EXECUTED graph only: 28 decoder layers + text "
                                      "head (the reference formula, kimi_audio/__init__.py:63-80, also counts the 6 mimo "
                                      "layers and the audio head, which a text-head step never runs), times this GPU's "
                                      "share of the tokens")
        if (args.cp > 1 or args.tp > 1):
            line["kernels_note"] = "per-kernel rooflines are reported on the headline workload (python bench.py)"
            try:                                   # (never lose the headline number to a diagnostics failure)
                rows_local = wl.B * wl.T // layout["cp"]
                lm_rows = rows_local
                toks = getattr(wl, "tokens", {})       # (the frontend-fed ASR workloads build their batch per step)
                if wl.job.training_enable_fused_ce and not args.all_rows_lm_head and not args.compact_lm_head:
                    bound = (toks.get("labelled_rows_max_cp", [None] * layout["cp"])[trainer.cp.rank]
                             if layout["cp"] > 1 else toks.get("labelled_rows_max"))
                    if bound is not None:
                        lm_rows = min(rows_local, (int(bound) + 255) // 256 * 256)
                line["config"]["lm_head_rows"] = f"{lm_rows} of this rank's {rows_local} positions"
                ex = executed_flops_per_gpu(wl, trainer, layout, lm_rows) / (elapsed / args.steps) / MFMA_PEAK
                line["step_mfu_executed_flops"] = round(ex, 4)
                line["roofline"].update({"achieved": round(ex * MFMA_PEAK / 1e12, 1), "frac": round(ex, 4),
                                         "formula_achieved": round(fpt * (tps / gpus_in_job) / 1e12, 1),
                                         "formula_frac": round(mfu, 4),
                                         "note": "one GPU's training step against the dense bf16 MFMA peak: `frac` on the "
                                                 "FLOPs this rank executes (bench.executed_flops_per_gpu), `formula_frac` by "
                                                 "the reference MFU formula on its share of the tokens"})
            except Exception as e:
                line["executed_flops_error"] = repr(e)
        elif args.workload != "tiny":
            try:
                step_ms = elapsed / args.steps * 1e3
                if args.no_kernel_rooflines:
                    # (the executed-FLOP fraction needs the allowed (query, key) pairs only, not the microbenchmarks:
                    #  VERDICT r4 #12 — with this switch the lines used to print the FORMULA value under that name)
                    allowed_pairs = allowed_attention_pairs(wl)
                else:
                    line["kernels"], allowed_pairs = kernel_rooflines(wl)
                    for k in line["kernels"]:
                        k["share_of_step"] = round(k["ms"] * k["launches_per_step"] / step_ms, 4)
                    # the kernel with the LARGEST share of the step (`roofline` itself stays the whole-step MFU the
                    # metric is defined on)
                    line["roofline"]["dominant_kernel"] = max(line["kernels"], key=lambda k: k["share_of_step"])
                # utilisation on the FLOPs the step actually executes: the formula credits 12*L*H*Dh*T of attention per
                # token (a full T x T triangle) while packing executes only the per-document triangles, and it counts
                # the audio tower's parameters once per TEXT token although it runs on its own frames
                c = wl.seq_cfg
                dec_attn = 3.5 * 4.0 * c.head_dim * c.num_attention_heads * allowed_pairs * c.num_hidden_layers
                executed = (fpt - 12 * c.num_hidden_layers * c.num_attention_heads * c.head_dim * wl.T) * wl.B * wl.T + dec_attn
                rows_run = decoder_rows(wl)
                if rows_run < wl.B * wl.T:
                    # the decoder's linear layers run on the non-pad slots only (padding slots dropped from the row work)
                    dec_layer = (c.hidden_size * c.num_attention_heads * c.head_dim * 2
                                 + 2 * c.hidden_size * c.num_key_value_heads * c.head_dim + 3 * c.hidden_size * c.intermediate_size)
                    executed -= 6.0 * dec_layer * c.num_hidden_layers * (wl.B * wl.T - rows_run)
                    line["config"]["decoder_rows"] = (f"{rows_run} of {wl.B * wl.T}: the non-pad slots only (packer's count "
                                                      f"{wl.tokens.get('valid_rows_max')} rounded up to 256)")
                if wl.name == "qwen2_audio_7b":        # (the short-utterance workload reports the formula MFU only)
                    ac = wl.model_config.audio_config
                    tower_params = sum(p.numel() for p in trainer.model.audio_tower.parameters())
                    frames = wl.wav.shape[0] * 1500
                    executed += 6.0 * tower_params * (frames - wl.B * wl.T)          # tower runs on frames, not tokens
                    executed += 3.5 * 4.0 * 64 * ac.encoder_attention_heads * wl.wav.shape[0] * (1500 * 1501 // 2) * ac.encoder_layers
                if lrm is not None and wl.job.training_enable_fused_ce and not c.tie_word_embeddings:
                    rows = min(wl.B * wl.T, (int(lrm) + 255) // 256 * 256)           # lm_head + CE run on these rows only
                    executed -= 6.0 * c.vocab_size * c.hidden_size * (wl.B * wl.T - rows)
                    import touchnet_amd.models.llama.modeling_llama as _ml
                    if _ml.LAST_LAYER_LABELLED_ROWS and 2 * rows <= wl.B * wl.T:
                        # ... and so does everything behind the LAST layer's attention core (o_proj + MLP)
                        executed -= 6.0 * (c.hidden_size * c.num_attention_heads * c.head_dim
                                           + 3 * c.hidden_size * c.intermediate_size) * (rows_run - rows)
                ex_frac = executed / (step_ms * 1e-3) / MFMA_PEAK
                line["step_mfu_executed_flops"] = round(ex_frac, 4)
                # the roofline's primary number is the utilisation on EXECUTED flops; the reference-formula value (which
                # credits attention pairs the document mask never computes) stays beside it and in `step_mfu`
                line["roofline"].update({"achieved": round(ex_frac * MFMA_PEAK / 1e12, 1), "frac": round(ex_frac, 4),
                                         "formula_achieved": round(fpt * (tps / gpus_in_job) / 1e12, 1),
                                         "formula_frac": round(mfu, 4),
                                         "note": "whole training step per GPU against the dense bf16 MFMA peak: `frac` on the "
                                                 "FLOPs the step executes, `formula_frac` by the reference MFU formula; "
                                                 "per-kernel rooflines of the hand-written HIP kernels in `kernels`"})
                line["executed_flops_note"] = ("GEMM terms of the formula (6*N_wo_emb per token; tower on its 1500 frames "
                                               "per clip; an untied lm_head only on the rows it runs on) + attention on "
                                               "the allowed (query, key) pairs only, fwd + 2.5x bwd")
            except Exception as e:  # never lose the headline number to a diagnostics failure
                line["kernels_error"] = repr(e)
        # HBM traffic: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command (scripts/step_traffic.sh), kept only
        # while it describes the code that is running: the file carries the digest of the kernel sources and the GEMM
        # mode it was taken with, and a stale one is reported as such instead of as a number
        import glob as _glob
        cands = sorted(_glob.glob(os.path.join(ROOT, "profiles", f"r*_step_hbm_traffic_{wl.name}.json")))
        tfile = cands[-1] if cands else ""                      # (the newest round's measurement)
        if os.path.exists(tfile):
            t = json.load(open(tfile))
            stamp = os.path.join(ROOT, "touchnet_amd", "_lib", "build.stamp")
            digest = open(stamp).read().strip() if os.path.exists(stamp) else None
            mode = __import__("touchnet_amd.functional", fromlist=["x"]).LINEAR_GEMM
            if t.get("kernel_sources_digest") == digest and t.get("linear_gemm") == mode:
                line["roofline"]["traffic"] = t["hbm_bytes_per_step"]
                line["roofline"]["traffic_source"] = t["source"]
            else:
                line["roofline"]["traffic_source"] = ("stale: profiles/" + os.path.basename(tfile) + " was taken with other "
                                                      "kernel sources / GEMM mode; re-run scripts/step_traffic.sh")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl, n_params=int(trainer.num_params))
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
