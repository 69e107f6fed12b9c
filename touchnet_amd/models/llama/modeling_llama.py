"""Packed-sequence Llama / Qwen2 causal LM on the HIP kernels.

Parameter names and shapes are exactly Hugging Face's (`model.embed_tokens.weight`,
`model.layers.N.self_attn.q_proj.weight`, ..., `lm_head.weight`) so the reference's DCP / HF
converters and `get_num_params` (touchnet/models/llama/__init__.py:57-67) keep working.
The arithmetic follows transformers/models/llama/modeling_llama.py:53-67,113-160,174-176,243-324
(see oracle/nn.py for the line-by-line restatement used as checker), re-scheduled for MI355X:

  * Q/K/V stay in the GEMM output layout [B, T, heads, D]; no transposes, no repeat_kv
  * residual add is fused into the RMSNorm that follows it (one HBM round trip instead of three)
  * RoPE tables come from the packed `position_ids` once per forward
  * attention = LDS-tiled MFMA kernel with document-masked block sparsity; the packers'
    `attention_mask` (document ids) is turned into tile metadata once per forward
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from ..backend import ops
from .configuration import DecoderConfig


class RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x, residual=None):
        return ops().rms_norm(x, self.weight, self.variance_epsilon, residual=residual)


class RotaryEmbedding(nn.Module):
    def __init__(self, config: DecoderConfig):
        super().__init__()
        self.config = config
        self.attention_scaling = 1.0
        self.register_buffer("inv_freq", self.compute_inv_freq(config), persistent=False)

    @staticmethod
    def compute_inv_freq(config, device=None):
        return ops().rope_inv_freq(config.head_dim, config.rope_theta, config.rope_scaling, device=device)

    def _apply(self, fn, recurse=True):
        """The frequencies stay float32 whatever the module is cast to.  `model.to(torch.bfloat16)` (the single-GPU
        Trainer's mixed-precision cast, every `.to(bf16)` of a test) used to round them to 8 bits: at position 700 the
        fastest pairs were then rotated by up to ~2.7 rad more or less than the reference rotates them (which keeps
        inv_freq fp32: transformers' *RotaryEmbedding.forward, touchnet/models/llama/__init__.py:19-36) — found by
        tests/test_full_size_parity_gpu.py (hidden states 7 % off the oracle on ~790-token documents; the short
        fixtures sit at positions where the error is below bf16 rounding)."""
        super()._apply(fn, recurse)
        inv = self.inv_freq
        if inv.dtype != torch.float32 and not inv.is_meta:
            self.inv_freq = self.compute_inv_freq(self.config, device=inv.device)
        return self

    def forward(self, position_ids, dtype):
        return ops().rope_tables(position_ids, self.inv_freq, dtype, self.attention_scaling)


class Attention(nn.Module):
    def __init__(self, config: DecoderConfig):
        super().__init__()
        H, D = config.hidden_size, config.head_dim
        self.num_heads, self.num_kv_heads, self.head_dim = config.num_attention_heads, config.num_key_value_heads, D
        self.q_proj = nn.Linear(H, self.num_heads * D, bias=config.attention_bias)
        self.k_proj = nn.Linear(H, self.num_kv_heads * D, bias=config.attention_bias)
        self.v_proj = nn.Linear(H, self.num_kv_heads * D, bias=config.attention_bias)
        self.o_proj = nn.Linear(self.num_heads * D, H, bias=False)
        self.scaling = D ** -0.5

    def forward(self, x, cos, sin, mask, keep_rows=None, norm_src=None):
        """`keep_rows` (flat indices into B*T, last decoder layer only): the output projection runs on those rows only and
        the result is [1, len(keep_rows), H] — see DecoderModel.forward.  `norm_src`: x is the RMSNorm of that tensor
        (functional.norm_source; op-level selective activation checkpointing)."""
        B, T, _ = x.shape
        cp = getattr(mask, "cp", None)
        if cp is not None and cp.need is not None:
            return self._forward_context_parallel(x, cos, sin, mask, cp)
        # one autograd node for the three projections of x: their weight gradients run as ONE GEMM in the forward
        # layout (functional._LinearGroup); parameters keep the HF names and shapes
        kw = {"norm_src": norm_src} if norm_src is not None else {}
        # q and k come back ROTATED: the rotary embedding sits in the epilogue of their projections (functional.gemm_rope;
        # shapes the epilogue does not take are rotated behind the product — same bits)
        # ... and their gradients come back from the attention node rotated back (functional._AttentionRopeGrad: the
        # transposed rotation in the attention backward's epilogues) unless the keys travel between ranks first
        grad_in_attn = (cp is None and getattr(ops(), "ROPE_GRAD_IN_ATTENTION", False) and x.dtype == torch.bfloat16
                        and cos.dtype == torch.bfloat16)
        if grad_in_attn:
            kw["rope_grad_in_attention"] = True
        q, k, v = ops().linear_group(x, [(self.q_proj.weight, self.q_proj.bias), (self.k_proj.weight, self.k_proj.bias),
                                         (self.v_proj.weight, self.v_proj.bias)], rope=(cos, sin, self.head_dim, (0, 1)), **kw)
        q = q.view(B, T, self.num_heads, self.head_dim)
        k = k.view(B, T, self.num_kv_heads, self.head_dim)
        v = v.view(B, T, self.num_kv_heads, self.head_dim)
        if cp is not None:            # context parallel, all-gather rotate method (utils/context_parallel.py)
            a = ops().packed_attention_sharded(q, cp.gather_seq(k), cp.gather_seq(v), mask, cp.seq_shard(),
                                               self.scaling)
        else:
            a = ops().packed_attention(q, k, v, mask, self.scaling, **({"rope_grad": (cos, sin)} if grad_in_attn else {}))
        a = a.view(B, T, self.num_heads * self.head_dim)
        if keep_rows is not None:
            a = a.reshape(B * T, -1).index_select(0, keep_rows)[None]
        return ops().linear_group(a, [(self.o_proj.weight, None)])[0]


    def _forward_context_parallel(self, x, cos, sin, mask, cp):
        """Halo-exchange form: K/V first, their exchange is issued, the QUERY path (projection + RoPE) and then the
        attention over the rank's OWN chunks run while the remote chunks travel over xGMI; the compute stream waits for
        them only in front of the attention over the received chunks, and the two parts are merged by their LSE.  The
        backward returns the partial dK/dV under the query-path backward GEMMs (utils/context_parallel.py)."""
        from touchnet_amd.utils.context_parallel import exchange_kv
        B, T, _ = x.shape
        k, v = ops().linear_group(x, [(self.k_proj.weight, self.k_proj.bias), (self.v_proj.weight, self.v_proj.bias)])
        k = k.view(B, T, self.num_kv_heads, self.head_dim)
        v = v.view(B, T, self.num_kv_heads, self.head_dim)
        k, _ = ops().apply_rope(k, k.new_empty(B, T, 0, self.head_dim), cos, sin)       # rotate K alone
        exchange = exchange_kv(cp, k, v)
        q = ops().linear_group(x, [(self.q_proj.weight, self.q_proj.bias)])[0].view(B, T, self.num_heads, self.head_dim)
        q, _ = ops().apply_rope(q, q.new_empty(B, T, 0, self.head_dim), cos, sin)
        # own-chunk attention while the remote chunks still travel, then the received chunks, merged by LSE
        a = exchange.attend(q, mask, self.scaling)
        return ops().linear_group(a.view(B, T, self.num_heads * self.head_dim), [(self.o_proj.weight, None)])[0]


# TN_SKIP_PAD_ROWS=0: keep the padding slots of a packed batch in every row-wise kernel (what the reference computes).
SKIP_PAD_ROWS = os.environ.get("TN_SKIP_PAD_ROWS", "1") != "0"


# TN_LAST_LAYER_LABELLED_ROWS=0: select the labelled rows in front of lm_head only (the round-2a behaviour), not in front
# of the last layer's output projection.
LAST_LAYER_LABELLED_ROWS = os.environ.get("TN_LAST_LAYER_LABELLED_ROWS", "1") != "0"


class MLP(nn.Module):
    def __init__(self, config: DecoderConfig):
        super().__init__()
        H, I = config.hidden_size, config.intermediate_size
        self.gate_proj = nn.Linear(H, I, bias=False)
        self.up_proj = nn.Linear(H, I, bias=False)
        self.down_proj = nn.Linear(I, H, bias=False)

    def forward(self, x, norm_src=None):
        # one autograd node: gate + up + SwiGLU as one launch, down; the backward's five launches (functional._SwiGLUMLP)
        kw = {"norm_src": norm_src} if norm_src is not None else {}
        return ops().swiglu_mlp(x, self.gate_proj.weight, self.up_proj.weight, self.down_proj.weight, **kw)


class DecoderLayer(nn.Module):
    def __init__(self, config: DecoderConfig):
        super().__init__()
        self.self_attn = Attention(config)
        self.mlp = MLP(config)
        self.input_layernorm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, delta, residual, cos, sin, mask, keep_rows=None):
        """`delta` is the previous sub-layer's output still to be added to the `residual` stream.
        `keep_rows`: everything behind the attention core (o_proj, residual, norm, MLP) runs on those rows only."""
        # `_tn_recompute_rows` (set by parallelize.apply_ac for the reference's selective AC option "op"): the GEMM nodes
        # keep the residual stream instead of the norm outputs and recompute the row kernels (norms, SwiGLU product) in
        # their backward — functional.norm_source
        sac = getattr(self, "_tn_recompute_rows", False) and torch.is_grad_enabled()
        if residual is None:
            residual = delta
            x = self.input_layernorm(delta)
        else:
            x, residual = self.input_layernorm(delta, residual)
        src = ops().norm_source(residual, self.input_layernorm.weight, self.input_layernorm.variance_epsilon) if sac else None
        a = self.self_attn(x, cos, sin, mask, keep_rows, norm_src=src)
        if keep_rows is not None:
            residual = residual.reshape(-1, residual.shape[-1]).index_select(0, keep_rows)[None]
        x, residual = self.post_attention_layernorm(a, residual)
        src = (ops().norm_source(residual, self.post_attention_layernorm.weight, self.post_attention_layernorm.variance_epsilon)
               if sac else None)
        return self.mlp(x, norm_src=src), residual


class DecoderModel(nn.Module):
    def __init__(self, config: DecoderConfig):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([DecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.rotary_emb = RotaryEmbedding(config)

    @staticmethod
    def _drop_pad_rows(inputs_embeds, position_ids, doc, keep_rows, valid_rows_max):
        """The packed batch without its padding slots, as ONE row: -> (embeds [1, Mc, H], position ids [1, Mc], document
        ids [1, Mc], keep_rows in the new numbering, `rows` = the flat B*T position of every kept slot, overflow flag) or
        None when nothing would be saved.

        The packers fill rows greedily, so a batch ends every row with padding (Qwen2-Audio ASR: 3-4 % of B x T).  The
        reference runs every layer on those slots and then ignores them: flex_attention gives them no key and no output
        (document id 0), every other op acts per position, their labels are -100.  Here the non-pad slots are gathered
        (batch rows one behind the other; document ids made unique across batch rows so that the document mask keeps them
        apart), the row count is rounded up to the GEMM tile height with real padding slots (document id 0: same
        semantics), and all GEMMs / row kernels / attention tiles of the decoder run on Mc < B*T rows.  Results on the
        non-pad positions are what the full computation gives.  `valid_rows_max` is the packers' count of non-pad slots
        (a host int: static shapes, no synchronisation); a bound that is too small is reported through the returned
        flag and poisons the output."""
        B, T, H = inputs_embeds.shape
        Mc = min((int(valid_rows_max) + 255) // 256 * 256, B * T)
        if Mc + 256 > B * T:
            return None
        flat = doc.reshape(-1)
        is_pad = flat <= 0
        # non-pad positions first (in their order), padding slots behind them: a permutation, so no index repeats
        order = torch.sort(is_pad.to(torch.int8), stable=True).indices
        rows = order[:Mc]
        overflow = (~is_pad).sum() > Mc
        uniq = doc.to(torch.int64) + (torch.arange(B, device=doc.device, dtype=torch.int64) * (T + 1))[:, None]
        doc_c = torch.where(is_pad, torch.zeros_like(flat, dtype=torch.int64), uniq.reshape(-1)).index_select(0, rows)[None]
        emb_c = inputs_embeds.reshape(B * T, H).index_select(0, rows)[None]
        pos_c = position_ids.reshape(-1).index_select(0, rows)[None]
        if keep_rows is not None:
            inv = torch.zeros(B * T, dtype=torch.int64, device=rows.device)
            inv[rows] = torch.arange(Mc, device=rows.device)
            keep_rows = inv.index_select(0, keep_rows)
        return emb_c, pos_c, doc_c, keep_rows, rows, overflow

    def forward(self, input_ids=None, inputs_embeds=None, position_ids=None, attention_mask=None,
                context_parallel=None, keep_rows=None, valid_rows_max=None):
        """With `context_parallel` (utils.context_parallel.ContextParallel): input_ids / inputs_embeds /
        position_ids are this rank's sequence shard [B, T/cp], `attention_mask` stays the GLOBAL [B, T]
        document-id tensor (it is tiny and every rank needs the tile metadata of the keys it attends to).

        `keep_rows` (int64 [R], flat indices into B*T): return the hidden states of THOSE rows only, [1, R, H].  Positions
        mix only inside the attention core, so in the LAST layer everything behind it — output projection, residual,
        norm, MLP, final norm: 3/4 of that layer's GEMM work — is computed for the kept rows alone; all earlier layers and
        the last layer's q/k/v + attention see every row (they are the keys and values of later positions).  The caller
        keeps the rows that carry a label (ASR-SFT batches: ~5 % of the positions); results on those rows are what the
        full computation gives."""
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids)
        B, T, _ = inputs_embeds.shape
        if position_ids is None:
            position_ids = torch.arange(T, device=inputs_embeds.device).expand(B, T)
        sp = getattr(self, "_tn_sp", None)
        dropped = None
        if (SKIP_PAD_ROWS and valid_rows_max is not None and context_parallel is None and sp is None
                and isinstance(attention_mask, torch.Tensor) and attention_mask.dim() == 2
                and not attention_mask.dtype.is_floating_point):
            dropped = self._drop_pad_rows(inputs_embeds, position_ids, attention_mask, keep_rows, valid_rows_max)
        if dropped is not None:
            inputs_embeds, position_ids, attention_mask, keep_rows, kept, overflow = dropped
            full_shape = (B, T)
            B, T = 1, inputs_embeds.shape[1]
        cos, sin = self.rotary_emb(position_ids, inputs_embeds.dtype)
        mask = attention_mask
        if mask is None:                                   # plain causal (Qwen2-Audio training path)
            mask = ops().causal_mask(B, T, inputs_embeds.device)
        elif isinstance(mask, torch.Tensor):               # the packers' document ids, [B, T] ints
            mask = ops().build_packed_mask(mask)
        if context_parallel is not None:
            mask.cp = context_parallel
        # tensor parallelism with sequence parallelism (models/tensor_parallel.py): the residual stream is this rank's
        # T/tp rows between the blocks; attention and MLP gather / reduce-scatter around their own bodies
        if sp is not None and keep_rows is not None:
            raise RuntimeError("keep_rows is not available under sequence parallelism")
        delta, residual = (inputs_embeds if sp is None else sp.scatter(inputs_embeds)), None
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            if keep_rows is not None and i == last:
                delta, residual = layer(delta, residual, cos, sin, mask, keep_rows)
            else:
                delta, residual = layer(delta, residual, cos, sin, mask)
        h, _ = self.norm(delta, residual)
        if dropped is not None:
            # a bound below the real count dropped real tokens: NaN in every output (and through it every gradient)
            h = h * torch.where(overflow, float("nan"), 1.0).to(h.dtype)
            if keep_rows is None:                            # back to [B, T, H]; the dropped padding slots read 0
                out = h.new_zeros(full_shape[0] * full_shape[1], h.shape[-1])
                h = out.index_copy(0, kept, h[0]).view(*full_shape, -1)
        return h if sp is None else sp.gather(h)


class PackedCausalLM(nn.Module):
    """Drop-in for LlamaForCausalLM / Qwen2ForCausalLM on packed batches
    (forward(**batch) -> object with `.logits`, the contract of touchnet/bin/train.py:440-452)."""
    base_model_prefix = "model"
    config_class = DecoderConfig

    def __init__(self, config: DecoderConfig):
        super().__init__()
        self.config = config
        self.model = DecoderModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        if config.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight

    def tie_weights(self):
        """`to_empty()` from the meta device re-creates every Parameter, which silently un-ties
        lm_head / embed_tokens; re-tie (HF does the same in its post-load hook)."""
        if self.config.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight

    def post_init(self):
        """HF-style init (normal(0, initializer_range) for Linear/Embedding, ones for norms)."""
        self.tie_weights()
        std = self.config.initializer_range
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0.0, std=std)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, mean=0.0, std=std)
            elif isinstance(m, RMSNorm):
                nn.init.ones_(m.weight)

    def _forward_labelled_rows(self, input_ids, inputs_embeds, position_ids, attention_mask, labels, sentence_lens,
                               num_sentence, ce_chunk_tokens, rows_max, ignore_index=-100, valid_rows_max=None):
        """Fused lm_head + CE when the packers supply an upper bound of the labelled positions: the rows are selected
        BEFORE the last layer's output projection instead of in front of lm_head (DecoderModel.forward, `keep_rows`).
        Static shapes, no host sync: the row list has round_up(rows_max, 256) entries, the filler entries repeat row 0
        with the label ignore_index (exact zeros in every gradient), and a bound that is too small poisons the loss with
        NaN on the device — the semantics of functional._FusedLinearCE's static compaction."""
        from touchnet_amd.loss.cross_entropy import fused_linear_cross_entropy
        lab = labels.reshape(-1).to(torch.int64)
        n_max = min((rows_max + 255) // 256 * 256, lab.numel())
        labelled = lab != ignore_index
        rows = torch.nonzero_static(labelled, size=max(n_max, 1), fill_value=0).squeeze(1)
        count = labelled.sum()
        valid = torch.arange(rows.numel(), device=lab.device) < count
        h = self.model(input_ids=input_ids, inputs_embeds=inputs_embeds, position_ids=position_ids,
                       attention_mask=attention_mask, keep_rows=rows, valid_rows_max=valid_rows_max)      # [1, R, H]
        lab_c = torch.where(valid, lab.index_select(0, rows), torch.full_like(rows, ignore_index))[None]
        sl_c = sentence_lens.reshape(-1).index_select(0, rows)[None]
        loss, per_token, acc = fused_linear_cross_entropy(h, self.lm_head.weight, lab_c, sl_c, num_sentence,
                                                          chunk_tokens=ce_chunk_tokens, compact=False)
        # A bound that is too small must not train on the truncated label set: poison the loss AND — through it — every
        # gradient (loss * NaN back-propagates NaN), so that the optimizer's device-side non-finite check skips the step;
        # `torch.where(overflow, nan, loss)` alone would give the real loss a ZERO gradient and AdamW would still apply
        # its weight decay and stale momentum.  The accuracy is poisoned as well.
        overflow = count > n_max
        poison = torch.where(overflow, float("nan"), 1.0).to(loss.dtype)
        loss = loss * poison
        per_token = per_token * poison.to(per_token.dtype)
        acc = acc * poison.to(acc.dtype) if isinstance(acc, torch.Tensor) else acc
        return SimpleNamespace(logits=None, loss=loss, loss_per_token=per_token, acc=acc)

    def forward(self, input_ids=None, inputs_embeds=None, position_ids=None, attention_mask=None,
                labels=None, sentence_lens=None, num_sentence=None, shift_labels=None,
                ce_chunk_tokens: int = 4096, ce_compact=False, labelled_rows_max=None, context_parallel=None,
                valid_rows_max=None, **unused):
        """Without `labels`: returns `.logits` (the reference's default path, loss_fn runs in the trainer).
        With `labels` (+ `sentence_lens`, `num_sentence`): lm_head and the packed CE run fused INSIDE the
        model — the role liger's fused-linear-CE plays in the reference (`pred.loss`, train.py:443-445), but
        with the per-sentence normalisation kept — and `.loss` / `.loss_per_token` / `.acc` are returned
        with `.logits = None`.  Being inside forward keeps lm_head under FSDP2's unshard/reshard hooks."""
        if (LAST_LAYER_LABELLED_ROWS and labels is not None and labelled_rows_max is not None and ce_compact is not True
                and context_parallel is None and len(self.model.layers) > 1 and getattr(self.model, "_tn_sp", None) is None
                and 2 * ((int(labelled_rows_max) + 255) // 256 * 256) <= labels.numel()):    # (pays for sparse labels only)
            return self._forward_labelled_rows(input_ids, inputs_embeds, position_ids, attention_mask, labels,
                                               sentence_lens, num_sentence, ce_chunk_tokens, int(labelled_rows_max),
                                               valid_rows_max=valid_rows_max)
        # (the padding slots are only skipped when the loss is formed in here: a caller that asks for logits gets every
        #  position computed as the reference computes it)
        h = self.model(input_ids=input_ids, inputs_embeds=inputs_embeds, position_ids=position_ids,
                       attention_mask=attention_mask, context_parallel=context_parallel,
                       valid_rows_max=valid_rows_max if (labels is not None or shift_labels is not None) else None)
        if labelled_rows_max is not None and ce_compact is not True:
            # the packers know how many positions carry a label: lm_head + CE run on those rows only, without a host
            # sync (functional._FusedLinearCE); rounded up so that the GEMM shapes repeat from step to step.  (Under
            # context parallelism the bound is the one of THIS rank's part of the labels — bin/train.py recounts it.)
            ce_compact = (int(labelled_rows_max) + 255) // 256 * 256
        if labels is None and shift_labels is not None:
            # The reference's liger branch (train.py:434-445 with training_enable_liger_kernel): the trainer pops
            # labels / sentence_lens / num_sentence and passes only `shift_labels`; `.loss` is then the MEAN over the
            # labelled tokens (liger's fused-linear-CE semantics, without the per-sentence normalisation).  Same fused
            # kernel path: every labelled token is its own "sentence", num_sentence = their count (device scalar).
            from touchnet_amd.loss.cross_entropy import fused_linear_cross_entropy
            n_valid = (shift_labels != -100).sum().clamp_min(1).to(torch.float32)
            loss, per_token, acc = fused_linear_cross_entropy(h, self.lm_head.weight, shift_labels,
                                                              torch.ones_like(shift_labels), n_valid,
                                                              chunk_tokens=ce_chunk_tokens, compact=ce_compact)
            return SimpleNamespace(logits=None, loss=loss, loss_per_token=per_token, acc=acc)
        lp = getattr(self, "_tn_loss_parallel", None)       # (group, rank, tp): lm_head holds V/tp rows of the vocabulary
        if labels is None:
            if lp is not None:
                raise RuntimeError("loss parallel: the head is vocabulary-sharded, use the fused lm_head + CE (pass labels)")
            return SimpleNamespace(logits=self.lm_head(h), loss=None)
        from touchnet_amd.loss.cross_entropy import fused_linear_cross_entropy
        loss, per_token, acc = fused_linear_cross_entropy(h, self.lm_head.weight, labels, sentence_lens, num_sentence,
                                                          chunk_tokens=ce_chunk_tokens, compact=ce_compact, tp=lp)
        return SimpleNamespace(logits=None, loss=loss, loss_per_token=per_token, acc=acc)
