/* touchnet_amd — C ABI of the MI355X (gfx950) packed-sequence training kernels.
 *
 * One shared library, libtouchnet_amd.so (hipcc --offload-arch=gfx950), plain pointers and sizes,
 * no torch / C++ types.  This is the drop-in boundary: the reference is pure Python
 * (SURVEY.md §0 fact 1), so the binding a TouchNet maintainer adds is a ctypes stub — see
 * INTEGRATION.md and touchnet_amd/_C.py (the binding this repo ships).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless named host_*; buffers are dense row-major
 *   - `dtype`: 0 = float32, 1 = bfloat16 (raw uint16 bits)
 *   - `stream`: hipStream_t cast to void* (NULL = default stream); all entry points only ENQUEUE work
 *     on that stream, never synchronise, never allocate; safe under hipGraph capture and from
 *     autograd worker threads (no global mutable state)
 *   - return value: 0 on success, a positive hipError_t from the launch, or -22 (EINVAL) for
 *     unsupported arguments.  Callers must treat non-zero as fatal (the reference has no per-op
 *     recovery either: touchnet/bin/train.py:639-648)
 *   - ownership: the caller owns every buffer; outputs are fully overwritten unless stated
 */
#ifndef TOUCHNET_AMD_H_
#define TOUCHNET_AMD_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ---- build identification -------------------------------------------------------------------- */
const char* tn_version(void);   /* "touchnet_amd <ver> gfx950" */

/* ---- RMSNorm (+ fused residual add)  — replaces LlamaRMSNorm / Qwen2RMSNorm forward+backward
 *      transformers/models/llama/modeling_llama.py:62-67 and the residual adds at :306-324,
 *      swapped in the way liger does at touchnet/models/llama/__init__.py:11-15.
 * fwd:  h = x (+ res_in);  res_out = h (if res_in);  y = w * T(h * rsqrt(mean(h^2) + eps));  rstd[rows] fp32
 * bwd:  dh = rmsnorm'(dy) (+ dres);  dw[H]     workspace: tn_norm_bwd_workspace_floats(rows, H) floats */
int tn_norm_bwd_workspace_floats(int rows, int H);
int tn_rmsnorm_fwd(const void* x, const void* res_in, const void* w, void* y, void* res_out, float* rstd,
                   int rows, int H, float eps, int dtype, void* stream);
int tn_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, void* dh,
                   void* dw, float* workspace, int rows, int H, int dtype, void* stream);

/* ---- LayerNorm (+ fused residual add) — Whisper-style encoder layers of the Qwen2-Audio tower
 *      (touchnet/models/qwen2_audio/__init__.py:18-133 drives transformers' Qwen2AudioEncoderLayer). */
int tn_layernorm_fwd(const void* x, const void* res_in, const void* w, const void* b, void* y, void* res_out,
                     float* mean, float* rstd, int rows, int H, float eps, int dtype, void* stream);
int tn_layernorm_bwd(const void* dy, const void* h, const void* w, const float* mean, const float* rstd,
                     const void* dres, void* dh, void* dw, void* db, float* workspace, int rows, int H, int dtype,
                     void* stream);

/* ---- SwiGLU / GELU — transformers/models/llama/modeling_llama.py:174-176; encoder fc1 activation */
int tn_swiglu_fwd(const void* gate, const void* up, void* out, long long n, int dtype, void* stream);
int tn_swiglu_bwd(const void* dout, const void* gate, const void* up, void* dgate, void* dup, long long n,
                  int dtype, void* stream);
/* bf16 variants that ALSO write the transposed tensors the MLP's weight-gradient GEMMs consume (rows, cols multiples
 * of 8): out_t [cols, rows] = out^T;  dgu_t [2*cols, rows] = [dgate^T ; dup^T].  Same arithmetic as the two above. */
int tn_swiglu_fwd_t(const void* gate, const void* up, void* out, void* out_t, int rows, int cols, void* stream);
int tn_swiglu_bwd_t(const void* dout, const void* gate, const void* up, void* dgate, void* dup, void* dgu_t, int rows,
                    int cols, void* stream);
int tn_gelu_fwd(const void* x, void* out, long long n, int dtype, void* stream);
int tn_gelu_bwd(const void* dout, const void* x, void* dx, long long n, int dtype, void* stream);

/* ---- RoPE from packed position_ids — transformers/models/llama/modeling_llama.py:113-160 with
 *      position_ids restarting per sentence (touchnet/models/llama/processing_llama.py:96-97).
 * table: cos/sin [n, half] in `dtype` = cos/sin(position_ids[n] * inv_freq[half]) * attention_scaling
 * apply: q [n, hq, D] -> q_out, k [n, hk, D] -> k_out (may alias; half-split rotate_half convention);
 *        backward = 1 applies the transpose rotation to gradients */
int tn_rope_table(const long long* position_ids, const float* inv_freq, void* cos_t, void* sin_t, int n, int half,
                  float attention_scaling, int dtype, void* stream);
int tn_rope_apply(const void* q, const void* k, void* q_out, void* k_out, const void* cos_t, const void* sin_t,
                  int n, int hq, int hk, int D, int backward, int dtype, void* stream);

/* ---- packed cross-entropy + accuracy — touchnet/loss/__init__.py:7-28,
 *      touchnet/loss/cross_entropy.py:12-50, touchnet/utils/metrics.py:26-50.
 * logits [n, V]; labels, sentence_lens int64 [n]; num_sentence, grad_out: device float[1]
 * forward: nll[n], lse[n] fp32, hit[n] int32, out[4] = {loss_per_sample, loss_per_token, accuracy, n_valid}
 * backward: dlogits (may alias logits) = (softmax - onehot) * grad_out / (sentence_lens * num_sentence) */
int tn_ce_forward(const void* logits, const long long* labels, const long long* sentence_lens,
                  const float* num_sentence, float* nll, float* lse, int* hit, float* out, int n, int V,
                  long long ignore_index, int dtype, void* stream);
int tn_ce_reduce(const float* nll, const int* hit, const long long* labels, const long long* sentence_lens,
                 const float* num_sentence, float* out, int n, long long ignore_index, void* stream);
int tn_ce_backward(const void* logits, void* dlogits, const long long* labels, const long long* sentence_lens,
                   const float* lse, const float* num_sentence, const float* grad_out, int n, int V,
                   long long ignore_index, int dtype, void* stream);

/* ---- packed (document-masked causal) attention — replaces flex_attention / SDPA
 *      (transformers/integrations/flex_attention.py:190-201,264-340 as selected by
 *      "attn_implementation": "flex_attention", examples/text/pretrain/fineweb-edu/config/Llama-3_2-1B.json:7;
 *      mask built from the packers' document ids, touchnet/models/llama/processing_llama.py:38-40).
 * q [B,T,Nh,D], k/v [B,T,Nkv,D], o [B,T,Nh,D] bf16, D in {64,128}; doc int32 [B,T] (0 = pad);
 * lse2, delta fp32 [B,Nh,T]; meta: int32[tn_attn_meta_ints(B,T)] filled by tn_attn_build_meta once per batch */
int tn_attn_meta_ints(int B, int T);
int tn_attn_build_meta(const int* doc, int* meta, int B, int T, void* stream);
int tn_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc, const int* meta,
                int B, int T, int Nh, int Nkv, int D, float scale, void* stream);
int tn_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse2,
                float* delta, void* dq, void* dk, void* dv, const int* doc, const int* meta, int B, int T, int Nh,
                int Nkv, int D, float scale, void* stream);
/* tn_attn_bwd for q / k that carry the rotary embedding (tn_gemm_bf16_rope's epilogue or tn_rope_apply put it there): dq / dk
 * are returned as gradients of the UN-rotated projections — tn_attn_bwd followed by tn_rope_apply(dq, dk, backward = 1), bit
 * for bit, with the transposed rotation done in the backward kernels' epilogues (D = 128) instead of one more pass over
 * dq / dk.  Replaces the autograd backward of transformers' apply_rotary_pos_emb inside the reference's decoder layers
 * (modeling_qwen2.py Qwen2Attention.forward as patched by touchnet/models/llama/__init__.py:11-15; flex_attention's
 * backward at touchnet/models/llama/parallelize_llama.py).  cos_t / sin_t: bf16 [B * T, D / 2] from tn_rope_table. */
int tn_attn_bwd_rope(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse2,
                     float* delta, void* dq, void* dk, void* dv, const int* doc, const int* meta, int B, int T, int Nh,
                     int Nkv, int D, float scale, const void* cos_t, const void* sin_t, void* stream);
/* The same pair with the mask BIDIRECTIONAL inside a document (allowed = same positive document id; no causal term):
 * transformers' WhisperEncoder layers, which the reference's Kimi-Audio speech encoder runs on every 30 s clip
 * (touchnet/models/kimi_audio/modeling_kimi_audio.py:933-960; one clip = one document).  Same tensors, same metadata. */
int tn_attn_fwd_bidir(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc, const int* meta,
                      int B, int T, int Nh, int Nkv, int D, float scale, void* stream);
int tn_attn_bwd_bidir(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse2,
                      float* delta, void* dq, void* dk, void* dv, const int* doc, const int* meta, int B, int T, int Nh,
                      int Nkv, int D, float scale, void* stream);

/* Sequence-sharded query side — context parallelism (replaces torch's experimental ring-attention CP that the
 * reference enters at touchnet/utils/distributed.py:292-346 / touchnet/bin/train.py:354-389, which intercepts
 * SDPA only and cannot carry the document mask).  q/o/dout/dq are [B, rows_per_batch, Nh, D] and lse2/delta
 * [B, Nh, rows_per_batch]: segment i = local rows [row0_i, row0_i+rows_i) holding GLOBAL positions
 * [off_i, off_i+rows_i) (head/tail load balancing = 2 segments per rank); `segs` = host int[6]
 * {row0_a, rows_a, off_a, row0_b, rows_b, off_b}.  k/v/doc/meta are global ([B, T, ...], all-gathered by the
 * caller); dk/dv come back as this rank's partial sums over [B, T, Nkv, D] for the caller to reduce-scatter. */
int tn_attn_fwd_seg(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                    const int* meta, int B, int T, int Nh, int Nkv, int D, float scale, int nseg,
                    const int* host_segs, int rows_per_batch, void* stream);
/*      Key side restricted to the sequence chunks whose bit is set in `chunk_mask` (chunk c = positions
 *      [c*chunk_len, (c+1)*chunk_len), chunk_len % 64 == 0, at most 64 chunks): rows that see no key there get
 *      O = 0, LSE2 = +inf.  With tn_attn_merge this is the local / remote split of context-parallel attention: the
 *      own-chunk part runs while the halo travels (replaces the per-step (out, lse) merge of torch's ring attention,
 *      torch/distributed/tensor/experimental/_context_parallel/_attention.py:182-183 entered from
 *      touchnet/utils/distributed.py:292-315). */
int tn_attn_fwd_seg_chunks(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                           const int* meta, int B, int T, int Nh, int Nkv, int D, float scale, int nseg, const int* segs,
                           int rows_per_batch, int chunk_len, unsigned long long chunk_mask, void* stream);
/*      lse = log2(2^lse_a + 2^lse_b), O = (2^lse_a O_a + 2^lse_b O_b) / 2^lse for two partial results over DISJOINT key
 *      sets; o may alias o_a, lse2 may alias lse2_a.  o_* [B, rows, Nh, D] bf16, lse2_* [B, Nh, rows] fp32. */
int tn_attn_merge(const void* o_a, const float* lse2_a, const void* o_b, const float* lse2_b, void* o, float* lse2,
                  int B, int rows, int Nh, int D, void* stream);
int tn_attn_bwd_seg(const void* q, const void* k, const void* v, const void* o, const void* dout,
                    const float* lse2, float* delta, void* dq, void* dk, void* dv, const int* doc, const int* meta,
                    int B, int T, int Nh, int Nkv, int D, float scale, int nseg, const int* host_segs,
                    int rows_per_batch, void* stream);

/* ---- audio frontend on device — touchnet/data/functions.py:117-134 (kaldi fbank),
 *      :159-190 (whisper log-mel), :258-286 (stack / stride / normalise).
 * fbank:  wav fp32 [n_samples] in [-1,1) -> feat fp32 [tn_fbank_frames(n_samples), n_mels]
 *         (x32768, 25 ms / 10 ms frames at 16 kHz, dc removal, pre-emphasis 0.97, povey window, 512-pt
 *         power spectrum, kaldi mel (20 Hz..Nyquist), log(max(., FLT_EPSILON)))
 * logmel: wav fp32 [n_samples] -> feat fp32 [n_samples/160, n_mels]; `mel_fb` = slaney filter bank
 *         fp32 [n_mels, 201] supplied by the host (it is a constant table)
 * stack:  feat fp32 [T, F] -> out fp32 [ceil(T/stride), F*stack], optional per-row normalisation */
int tn_fbank_frames(int n_samples);
int tn_kaldi_fbank(const float* wav, float* feat, int n_samples, int n_mels, void* stream);
int tn_log_mel(const float* wav, const float* mel_fb, float* feat, float* scratch_max, int n_samples, int n_mels,
               void* stream);
int tn_audiofeat_stack(const float* feat, float* out, int T, int F, int stack, int stride, int normalize,
                       void* stream);
/* Feature-level augmentation of the reference's datapipe, touchnet/data/functions.py:193-255 (audiofeat_spec_aug,
 * audiofeat_spec_sub, audiofeat_spec_trim), applied in ONE gather pass: x [T, F] -> y [out_rows, F].  The random draws
 * are the caller's (host arrays): t_masks / f_masks = n x [start, end) stripes set to zero, subs = n x (start, end, pos):
 * rows [start, end) take rows [start - pos, end - pos) of x (later entries win), out_rows <= T drops the tail.  At most 16
 * entries each; y must not alias x. */
int tn_feat_augment(const float* x, float* y, int T, int F, int out_rows, const int* t_masks, int n_t,
                    const int* f_masks, int n_f, const int* subs, int n_sub, void* stream);
/* Waveform-level speed perturbation, touchnet/data/functions.py:99-114 (sox `speed s` + `rate`): x [n_in] resampled to
 * y [n_out], output n at input position n p / q, by a polyphase band-limited interpolation table tab [q][ntap] (device,
 * float; built by the host side, touchnet_amd/functional.py::speed_perturb).  Not bit-comparable with libsox. */
int tn_resample_polyphase(const float* x, float* y, const float* tab, long long n_in, long long n_out, int p, int q,
                          int ntap, void* stream);

/* ---- fused AdamW on fp32 master weights with bf16 shadow write-back and device-side
 *      skip-on-nonfinite — touchnet/utils/optimizer.py:157-172 + touchnet/bin/train.py:458-474.
 * step 1: tn_sumsq accumulates sum(g^2) of one (flat) tensor into norm_sq[0] (fp32, zeroed by the caller);
 *         deterministic two-stage reduction through `scratch` (tn_sumsq_scratch_floats() floats)
 * step 2: tn_adamw_step updates p/m/v (fp32) from g, scaling g by min(1, max_norm/(sqrt(norm_sq)+1e-6));
 *         if norm_sq is NaN/Inf nothing is written (the reference skips the step, train.py:467-473) */
int tn_sumsq_scratch_floats(void);
int tn_sumsq(const void* g, float* scratch, float* norm_sq, long long n, int dtype, void* stream);
/* multi-tensor step 1 (one launch for all gradient tensors of one dtype): device tables ptrs[t], sizes[t]
 * (elements), first_chunk[t] = sum_{u<t} ceil(sizes[u] / tn_sumsq_multi_chunk()); partial: nchunks floats */
long long tn_sumsq_multi_chunk(void);
int tn_sumsq_multi(const void* const* ptrs, const long long* sizes, const long long* first_chunk, int ntensors,
                   long long nchunks, float* partial, float* norm_sq, int dtype, void* stream);
int tn_adamw_step(float* p, float* m, float* v, const void* g, void* p_shadow_bf16, const float* norm_sq,
                  long long n, float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                  float bias_corr1, float bias_corr2, int g_dtype, void* stream);

/* multi-tensor step 2 (one launch per gradient dtype for ALL parameters — FSDP shards are 1/N of each tensor, a launch
 * per tensor would dominate).  tn_adamw_prepare turns norm_sq into the device-side step state (8 floats owned by the
 * caller, zeroed once): step count — advanced only when the norm is finite, like torch's AdamW on a skipped step —
 * bias corrections, clip coefficient, skip flag; no host round trip.  Tables as in tn_sumsq_multi with
 * first_chunk over tn_adamw_multi_chunk(); shadows[t] may be NULL. */
long long tn_adamw_multi_chunk(void);
int tn_adamw_prepare(const float* norm_sq, float* state, float beta1, float beta2, float max_norm, void* stream);
int tn_adamw_multi(void* const* ps, void* const* ms, void* const* vs, const void* const* gs, void* const* shadows,
                   const long long* sizes, const long long* first_chunk, int ntensors, long long nchunks,
                   const float* state, float lr, float beta1, float beta2, float eps, float weight_decay, int g_dtype,
                   void* stream);
/* The same update from at most `max_workgroups` workgroups (0 = one per chunk = tn_adamw_multi), each striding over the
 * chunks: the form for a SIDE stream, where the update runs beside the next forward's kernels and must leave them the
 * machine — the unbounded launch takes the whole HBM bandwidth and the MFMA kernels beside it run at half speed
 * (profiles/r04e_*).  Same arithmetic per element. */
int tn_adamw_multi_bounded(void* const* ps, void* const* ms, void* const* vs, const void* const* gs, void* const* shadows,
                           const long long* sizes, const long long* first_chunk, int ntensors, long long nchunks,
                           const float* state, float lr, float beta1, float beta2, float eps, float weight_decay,
                           int g_dtype, int max_workgroups, void* stream);

/* ---- bf16 transpose for the weight-gradient GEMMs of the linear layers: dst[c*dst_ld + r] = src[r*src_ld + c].
 *      Replaces the implicit operand transposes of torch.nn.functional.linear's backward (every nn.Linear of the
 *      decoder blocks the reference trains, touchnet/bin/train.py:440-470): dW = dY^T X is run with BOTH operands
 *      contraction-contiguous (the forward GEMM's layout), which hipBLASLt executes ~1.4x faster on MI355X.
 *      rows, cols, leading dimensions: multiples of 8; base addresses 16-byte aligned (else -22). */
int tn_transpose_bf16(const void* src, void* dst, int rows, int cols, long long src_ld, long long dst_ld,
                      void* stream);

/* ---- bias gradient of a linear layer: out[c] = sum_r x[r*ld + c], x bf16 [rows, cols] (cols, ld multiples of 8),
 *      out bf16 [cols]; fp32 accumulation, deterministic two-stage reduction through ws
 *      (tn_colsum_workspace_floats(rows, cols) floats).  Replaces the `grad_output.sum(0)` of
 *      torch.nn.functional.linear's backward for the biased projections (Qwen2 q/k/v, the audio tower). */
long long tn_colsum_workspace_floats(int rows, int cols);
int tn_colsum_bf16(const void* x, void* out, float* ws, int rows, int cols, long long ld, void* stream);

/* ---- int16 PCM of a TouchDataset audio shard -> float32 [-1, 1) on the device (x / 32768, exact): the conversion
 *      the reference does on a CPU worker, touchnet/data/datapipe.py:163-165 */
int tn_pcm16_to_f32(const void* pcm_int16, float* out, long long n, void* stream);

/* ---- BEST-RQ labels: codes[t] = argmin_v || normalize(feat[t] @ quantizer) - codebook[v] ||_2 —
 *      BestRQTokenizer.tokenize, touchnet/tokenizer/tokenizer.py:289-299 (called per utterance from
 *      batch_audio_packed, touchnet/models/touch_audio/processing_touch_audio.py:97).
 *      feat [T, F], quantizer [F, E], codebook [V, E] (already L2-normalised, 16-byte aligned) fp32; codes int64 [T];
 *      E in {8, 16, 32} (else -22).  First index wins ties, like torch.argmin. */
int tn_bestrq_tokenize(const float* feat, const float* quantizer, const float* codebook, long long* codes, int T,
                       int F, int E, int V, void* stream);

/* ---- greedy first-fit sequence packing on the device: the placement + scatter of
 *      touchnet/models/llama/processing_llama.py:24-104 (batch_text) and
 *      touchnet/models/touch_audio/processing_touch_audio.py:117-214 (batch_pairaudio_pairtext_packed), bit-identical.
 * plan: lens int32 [n] (slots per segment: tokens + 1, plus feature rows for audio+text) -> row / col / sent (document
 *       index inside the row, from 1) / batch int32 [n]; counts int32 [3] = {batches with >= 1 segment, segments longer
 *       than T, error flag (set when such a segment exists and skip_too_long == 0)}; skipped segments get row = -1.
 * fill: writes the five int64 [B, T] buffers (pad / -100 / 0 / 0 / 1 elsewhere) and, when feats != NULL, the fp32
 *       [B, T, F] feature buffer of batch `which_batch`; segment i = alen[i] feature rows then bos + ntok[i] tokens
 *       (labels pre-shifted: tokens + eos); tok_off / feat_off: offsets of segment i in `tokens` / `feat_src` rows. */
int tn_pack_plan(const int* lens, int n, int B, int T, int skip_too_long, int* row, int* col, int* sent, int* batch,
                 int* counts, void* stream);
int tn_pack_fill(const int* row, const int* col, const int* sent, const int* batch, int which_batch, int n,
                 const int* ntok, const long long* tok_off, const long long* tokens, const int* alen,
                 const long long* feat_off, const float* feat_src, int F, int B, int T, long long bos, long long eos,
                 long long pad, long long* input_ids, long long* labels, long long* position_ids, long long* doc,
                 long long* sentence_lens, float* feats, int* nsent, void* stream);

/* ---- hand-written bf16 MFMA GEMM for the linear layers (q/k/v/o/gate/up/down/lm_head of the decoder blocks,
 *      fc1/fc2/out_proj of the audio tower): replaces torch.nn.functional.linear / liger's MLP swap,
 *      touchnet/models/llama/__init__.py:11-15 — forward, input-gradient and weight-gradient
 *      products alike, each reading nn.Linear's own tensors (no transposed copies):
 *        C[M,N] = sum_{s < nseg} opA_s . opB_s^T (+ bias[N]) (+ C when accumulate != 0)
 *      bf16 in/out, fp32 accumulation over ALL segments, ONE rounding.
 *      a_kmaj / b_kmaj = 0: the operand is stored [rows, K] (contraction-contiguous: x and W in y = x W^T, dY in
 *      dX = dY W); = 1: stored [K, rows] (contraction-major: W in dX = dY W; dY and x in dW = dY^T x).
 *      (a_kmaj = 1, b_kmaj = 0) has no caller and returns -22.  A, B, lda, ldb, K are arrays of nseg (1..3) entries;
 *      Ct (optional, may be NULL): transposed copy [N, M] written by the same epilogue.
 *      -22 unless: every K % 64 == 0 (any K > 0 when both operands are contraction-major), N % 8 == 0, every ld % 8 == 0 and >= the operand's contiguous extent, 16-byte
 *      aligned bases; row-stored operands 288 * ld * 2 < 2^31, contraction-major operands ((K-1) * ld + rows) * 2 < 2^31;
 *      a_kmaj: M % 8 == 0; with Ct: M % 8 == 0, ldct % 8 == 0, accumulate == 0. */
int tn_gemm_bf16(const void* const* A, const void* const* B, const long long* lda, const long long* ldb, const int* K,
                 int nseg, int a_kmaj, int b_kmaj, void* C, void* Ct, const void* bias, int M, int N, long long ldc,
                 long long ldct, int accumulate, void* stream);
/*      C = A B^T (+ bias) + addend: the residual stream added in the epilogue of the producing GEMM — `hidden_states =
 *      residual + hidden_states` behind o_proj / down_proj (transformers' LlamaDecoderLayer / Qwen2DecoderLayer.forward as
 *      driven by touchnet/models/llama/__init__.py; WhisperEncoderLayer.forward for the audio tower) — so that the norm
 *      behind it reads one tensor instead of two.  One segment; addend [M, N] bf16 with pitch ldadd (>= N, % 8), 16-byte
 *      aligned, must not alias C; the product is rounded to bf16 before the addition (the bits of GEMM + residual add). */
int tn_gemm_bf16_addend(const void* A, const void* B, long long lda, long long ldb, int K, int a_kmaj, int b_kmaj, void* C,
                        const void* bias, const void* addend, long long ldadd, int M, int N, long long ldc, void* stream);
/*      The same product (one segment, no transposed copy) with the contraction cut into `splitk` >= 2 parts that run as
 *      independent units.  tail_only = 0: every tile is split — outputs of few 256 x 256 tiles with a deep contraction (the
 *      audio tower's 1280 x 1280 weight gradients contract 30000 frames: 25 tiles for 256 CUs).  tail_only = 1: the whole
 *      rounds of tiles run unsplit and only the last partial round is split (the MLP weight gradients are 688 tiles = 2
 *      rounds + 176: cut in 4 the remainder costs 0.75 of a round instead of 1).  fp32 partial sums travel through
 *      `workspace` (whole-tile slabs: >= splitk * split_tiles * 262144 bytes, 16-byte aligned); a second kernel adds them
 *      (+ bias, + C when accumulate).  -22 additionally when an operand is contraction-contiguous and
 *      ceil(K / 64) % splitk != 0, or tail_only without a whole round / without a remainder. */
int tn_gemm_bf16_splitk(const void* A, const void* B, long long lda, long long ldb, int K, int a_kmaj, int b_kmaj,
                        void* C, const void* bias, int M, int N, long long ldc, int accumulate, int splitk, int tail_only,
                        void* workspace, long long workspace_bytes, void* stream);
/*      Weight gradient with FP32 output — for a data-parallel engine that reduces gradients in fp32 (the reference's
 *      MixedPrecisionPolicy(reduce_dtype=float32), touchnet/models/helper_func.py:165) and wants dW written straight into its
 *      reduce-scatter input instead of cast-copied there: C[M,N] (float, ldc in floats) = (+= when accumulate)
 *      A[K,M]^T . B[K,N], both operands contraction-major as stored (dY [tokens, M], x [tokens, N]).  splitk <= 1: one
 *      launch; >= 2: split-K through `workspace` (>= splitk * tiles * 262144 bytes).  Same -22 rules as tn_gemm_bf16. */
int tn_gemm_bf16_wgrad_f32(const void* A, const void* B, long long lda, long long ldb, int K, float* C, int M, int N,
                           long long ldc, int accumulate, int splitk, void* workspace, long long workspace_bytes,
                           void* stream);
/*      Weight gradient AND bias gradient of y = x W^T + b (nn.Linear with bias: Qwen2's q/k/v projections,
 *      transformers' Qwen2Attention behind touchnet/models/qwen2_audio/__init__.py; every Whisper-tower layer) in ONE launch:
 *      C[M, N] (bf16; float with ldc in floats when c_f32; += when accumulate) = A[K, M]^T . B[K, N], bias_grad[M] (bf16)
 *      = column sums of A (A = dY [tokens, M], B = x [tokens, N] as stored).  The sums are taken from the dY fragments
 *      the matrix pipe reads anyway: the separate column-sum pass (tn_colsum_bf16, one more trip of dY through HBM) is
 *      not needed.  splitk >= 2: split-K, `workspace` >= (splitk * tiles * 65536 + splitk * ceil(M / 256) * 256) * 4 bytes.
 *      -22 as for tn_gemm_bf16 with both operands contraction-major; bias_grad must not be NULL. */
int tn_gemm_bf16_wgrad_bias(const void* A, const void* B, long long lda, long long ldb, int K, void* C, void* bias_grad,
                            int M, int N, long long ldc, int accumulate, int c_f32, int splitk, void* workspace,
                            long long workspace_bytes, void* stream);
/*      Several INDEPENDENT products of one operand mode as one persistent launch (ngrp = 1..3, weight-gradient mode
 *      a_kmaj = b_kmaj = 1 only): C_g[M_g, N_g] (+= when accumulate) A_g[K_g, M_g]^T . B_g[K_g, N_g], bf16 outputs (ldc in
 *      elements) or, c_f32 = 1, float outputs (ldc in floats).  The three weight gradients of transformers' LlamaMLP
 *      (gate / up / down_proj, swapped at touchnet/models/llama/__init__.py:11-15) are 688 output tiles each — 2.69
 *      rounds on 256 CUs; as ONE tile list they are 8 whole rounds + 16 tiles, and that remainder (tiles mod CUs, when at
 *      most half the CUs and every contraction >= 16 stages) runs split-K through `workspace`
 *      (>= tn_gemm_grouped_workspace_bytes(M, N, K, ngrp) bytes, 16-byte aligned; NULL / too small: the remainder runs as
 *      whole tiles).  -22: shapes as for tn_gemm_bf16 with both operands contraction-major. */
long long tn_gemm_grouped_workspace_bytes(const int* M, const int* N, const int* K, int ngrp);
int tn_gemm_bf16_grouped(const void* const* A, const void* const* B, const long long* lda, const long long* ldb,
                         const int* K, void* const* C, const long long* ldc, const int* M, const int* N, int ngrp,
                         int a_kmaj, int b_kmaj, int accumulate, int c_f32, void* workspace, long long workspace_bytes,
                         void* stream);
/*      The MLP's gate and up products with SwiGLU in the epilogue (LlamaMLP.forward: down(silu(gate(x)) * up(x));
 *      the reference swaps in liger's fused SwiGLU at touchnet/models/llama/__init__.py:11-15): one launch computes
 *      gate[M, I] = x Wg^T, up[M, I] = x Wu^T and act = silu(gate) * up (all bf16, row pitch ldc; gate and up are kept
 *      for the backward) from x [M, K] (pitch ldx) and Wg, Wu [I, K] (pitch ldw) — an output tile is 256 rows x (128 gate
 *      + 128 up columns), no separate SwiGLU pass.  Arithmetic = tn_swiglu_fwd on the bf16-rounded products (bit-identical).
 *      -22 unless K % 64 == 0, I % 8 == 0, pitches % 8 == 0, 16-byte aligned bases. */
int tn_gemm_bf16_swiglu_fwd(const void* x, const void* wg, const void* wu, void* gate, void* up, void* act, int M, int I,
                            int K, long long ldx, long long ldw, long long ldc, void* stream);
/*      Its backward counterpart: d(act) = dY W_down (dY [M, H], pitch lddy; W_down [H, I] read contraction-major, pitch
 *      ldw) stays in the accumulators; the epilogue reads gate / up [M, I] and writes d(gate), d(up) (pitch ld) =
 *      tn_swiglu_bwd on the bf16-rounded d(act) (bit-identical).  -22 unless H % 64 == 0, I % 8 == 0, pitches % 8 == 0. */
int tn_gemm_bf16_swiglu_bwd(const void* dy, const void* wd, const void* gate, const void* up, void* dgate, void* dup,
                            int M, int I, int H, long long lddy, long long ldw, long long ld, void* stream);
/*      The audio tower's MLP (transformers' WhisperEncoderLayer.forward: fc2(gelu(fc1(x))), exact-erf GELU, as driven by
 *      touchnet/models/qwen2_audio/__init__.py:18-133): pre[M, N] = x W^T + bias AND act = gelu(pre) from one launch
 *      (bit-identical to tn_gemm_bf16 followed by tn_gelu_fwd on the rounded pre).  -22 unless K % 64 == 0, N % 8 == 0,
 *      pitches % 8 == 0 and >= the rows they span, 16-byte aligned bases, pre != act. */
int tn_gemm_bf16_gelu_fwd(const void* x, const void* w, const void* bias, void* pre, void* act, int M, int N, int K,
                          long long ldx, long long ldw, long long ldc, void* stream);
/*      ... and the backward of its second half: d(pre)[M, I] = (dY[M, H] W2[H, I]) o gelu'(pre): d(act) lives in the
 *      accumulators only (W2 read contraction-major, pitch ldw; pre / d(pre) with pitch ld) = tn_gemm_bf16 (b_kmaj) followed
 *      by tn_gelu_bwd on the rounded d(act) (bit-identical).  -22 unless H % 64 == 0, I % 8 == 0, pitches % 8 == 0. */
int tn_gemm_bf16_gelu_bwd(const void* dy, const void* w2, const void* pre, void* dpre, int M, int I, int H, long long lddy,
                          long long ldw, long long ld, void* stream);
/*      A q / k projection with the rotary embedding in the epilogue (transformers' LlamaAttention.forward: q_proj /
 *      k_proj followed by apply_rotary_pos_emb, modeling_llama.py:113-160): out[M, N] = rope(x W^T + bias), N = heads x
 *      head_dim (64 or 128), cos / sin [M, head_dim / 2] bf16 = one table row per output row (tn_rope_table).  Bit-identical
 *      to tn_gemm_bf16 followed by tn_rope_apply.  -22 unless K % 64 == 0, pitches % 8 == 0, N % 256 == 0 (head_dim 128) /
 *      N % 64 == 0 (head_dim 64). */
int tn_gemm_bf16_rope(const void* x, const void* w, const void* bias, const void* cos_t, const void* sin_t, void* out, int M,
                      int N, int K, long long ldx, long long ldw, long long ldc, int head_dim, void* stream);
/*      Single segment, both operands contraction-contiguous (the round-2 entry point): */
int tn_gemm_bf16_tn(const void* A, const void* B, void* C, void* Ct, const void* bias, int M, int N, int K,
                    long long lda, long long ldb, long long ldc, long long ldct, int accumulate, void* stream);

/*      Launch form of the GEMMs above: 1 (default) = persistent, one workgroup per CU walking a fixed tile list; 0 = one
 *      workgroup per tile, for a host that runs collectives beside the compute (an RCCL kernel holding CUs would leave a
 *      persistent workgroup's whole tile list waiting: the FSDP2 / context-parallel schedule of
 *      touchnet/models/helper_func.py:134-202, touchnet/utils/distributed.py:292-346).  Process-wide, takes effect at the
 *      next launch; TN_GEMM_PERSIST in the environment overrides it (kernel-development A/B).  Returns nothing / the value. */
void tn_gemm_set_persistent(int on);
int tn_gemm_get_persistent(void);

#ifdef __cplusplus
}
#endif
#endif /* TOUCHNET_AMD_H_ */
