"""The drop-in boundary against what the REFERENCE declares and calls: tests/golden/boundary.json holds its
TrainSpec field list, ParallelDims surface, apply_fsdp parameters and the way touchnet/bin/train.py calls every hook
(generated from the reference's syntax trees by tests/golden/make_golden.py::boundary_case — data only)."""
import dataclasses
import inspect
import json
import os
import types

import numpy as np
import pytest
import torch

import oracle.ops as oracle_ops
from touchnet_amd.models.backend import use_ops

HERE = os.path.dirname(os.path.abspath(__file__))
B = json.load(open(os.path.join(HERE, "golden", "boundary.json")))


def test_train_spec_schema_is_the_references():
    from touchnet_amd.utils.train_spec import TrainSpec
    fields = dataclasses.fields(TrainSpec)
    assert [f.name for f in fields] == B["train_spec_fields"]                       # names AND order
    required = [f.name for f in fields if f.default is dataclasses.MISSING and f.default_factory is dataclasses.MISSING]
    assert required == B["train_spec_required"]


def test_registry_errors_like_the_reference():
    from touchnet_amd.utils import train_spec as ts
    import touchnet_amd.specs  # noqa: F401
    with pytest.raises(ValueError, match="is not registered"):
        ts.get_train_spec("no_such_model")
    with pytest.raises(ValueError, match="already registered"):
        ts.register_train_spec(ts.get_train_spec("llama_mi355"))
    seen = []
    ts.apply_to_train_specs(lambda s: (seen.append(s.name), s)[1])
    assert {"llama_mi355", "touch_audio_mi355", "qwen2_audio_mi355"} <= set(seen)


@pytest.mark.parametrize("name", ["llama_mi355", "touch_audio_mi355", "qwen2_audio_mi355"])
def test_registered_hooks_accept_the_reference_trainers_calls(name):
    """Every call site of touchnet/bin/train.py must bind against the hook's signature: positional count, keywords."""
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.utils.train_spec import get_train_spec
    spec = get_train_spec(name)
    for hook, sites in B["train_py_calls"].items():
        fn = getattr(spec, hook)
        if hook == "pipelining_fn":
            assert fn is None                                                       # PP out of scope
            continue
        assert callable(fn), hook
        sig = inspect.signature(fn)
        for site in sites:
            args = [object()] * site["n_positional"]
            kwargs = {k: object() for k in site["keywords"]}
            if site["star_kwargs"]:
                kwargs["some_special_token"] = object()
            try:
                sig.bind(*args, **kwargs)
            except TypeError as e:
                raise AssertionError(f"{name}.{hook} cannot be called as train.py:{site['line']} does: {e}") from None


def test_integration_shim_builds_the_references_dataclass_from_our_specs():
    """`TrainSpec(**fields of our spec)` into a dataclass with the REFERENCE's field list (positional
    construction in its order works too)."""
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.utils.train_spec import get_train_spec
    RefSpec = dataclasses.make_dataclass("TrainSpec", [(n, object) for n in B["train_spec_fields"]])
    for name in ("llama_mi355", "touch_audio_mi355", "qwen2_audio_mi355"):
        ours = get_train_spec(name)
        kw = {f.name: getattr(ours, f.name) for f in dataclasses.fields(ours)}
        ref = RefSpec(**kw)
        assert ref.name == name and ref.parallelize_fn is ours.parallelize_fn
        assert RefSpec(*[kw[n] for n in B["train_spec_fields"]]) == ref


def test_parallel_dims_surface_and_mesh_names():
    from touchnet_amd.utils.distributed import ParallelDims
    assert [f.name for f in dataclasses.fields(ParallelDims)] == B["parallel_dims"]["fields"]
    for prop in B["parallel_dims"]["properties"]:
        assert isinstance(getattr(ParallelDims, prop), property), prop
    d = ParallelDims(dp_replicate=1, dp_shard=-1, cp=2, tp=1, pp=1, world_size=8, enable_loss_parallel=False)
    assert d.dp_shard == 4 and d.cp_enabled and d.dp_shard_enabled and not d.tp_enabled
    assert d.non_data_parallel_size == 2
    with pytest.raises(AssertionError):
        ParallelDims(dp_replicate=1, dp_shard=3, cp=2, tp=1, pp=1, world_size=8)
    src = inspect.getsource(ParallelDims.build_mesh)
    for nm in B["parallel_dims"]["mesh_dim_names"]:
        assert f'"{nm}"' in src, nm                                                 # same mesh-dimension names


def test_apply_fsdp_takes_the_references_leading_parameters():
    from touchnet_amd.models.helper_func import apply_fsdp
    ours = list(inspect.signature(apply_fsdp).parameters)
    assert ours == B["apply_fsdp_params"][:len(ours)]


def _tiny_cfg():
    from touchnet_amd.models.llama import DecoderConfig
    return DecoderConfig(vocab_size=32, hidden_size=32, intermediate_size=64, num_hidden_layers=4, num_attention_heads=4,
                         num_key_value_heads=2, rope_theta=10000.0, rms_norm_eps=1e-5,
                         tie_word_embeddings=False)


def test_parallelize_fn_four_positionals_on_meta_model_with_activation_checkpointing():
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.utils.distributed import ParallelDims
    from touchnet_amd.utils.train_spec import get_train_spec
    spec = get_train_spec("llama_mi355")
    dims = ParallelDims(1, 1, 1, 1, 1, 1, False)
    job = types.SimpleNamespace(training_activation_checkpoint_mode="selective",
                                training_activation_checkpoint_selective_ac_option="2", training_compile=False,
                                training_enable_cpu_offload=False)
    with torch.device("meta"):
        model = spec.model_cls(_tiny_cfg())
    out = spec.parallelize_fn(model, None, dims, job)                              # world_mesh unused without dp/cp/tp
    assert out is model
    wrapped = [type(b).__name__ == "CheckpointWrapper" for b in model.model.layers]
    assert wrapped == [False, True, False, True]                                    # every 2nd block
    job.training_activation_checkpoint_mode = "bogus"
    with pytest.raises(ValueError, match="Invalid AC mode"):
        spec.parallelize_fn(model, None, dims, job)
    with pytest.raises(NotImplementedError):
        spec.parallelize_fn(model, None, ParallelDims(1, 1, 1, 1, 2, 2, False), job)


def test_activation_checkpointing_does_not_change_loss_or_gradients():
    """Blocks re-executed in backward (full AC through the TrainSpec hook) == plain run, on the oracle op set."""
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.bin.train import TrainConfig, Trainer
    from touchnet_amd.data.synthetic import text_batch
    res = {}
    with use_ops(oracle_ops):
        for mode in ("none", "full", "op"):
            # "op" = the reference's op-level selective policy (helper_func.py:39-96): the hook marks the decoder blocks and
            # their GEMM nodes are handed the norm's input (`norm_source`); on the oracle op set that is wiring only
            job = TrainConfig(training_model_name="llama_mi355", training_enable_fused_ce=False,
                              training_activation_checkpoint_mode="selective" if mode == "op" else mode,
                              training_activation_checkpoint_selective_ac_option="op" if mode == "op" else "2")
            tr = Trainer(job, _tiny_cfg(), torch.device("cpu"),
                         optimizer_factory=lambda ps: torch.optim.SGD(ps, lr=0.0))
            if mode == "op":
                assert all(getattr(b, "_tn_recompute_rows", False) for b in tr.model.model.layers)
                seen = []
                inner = oracle_ops.norm_source
                oracle_ops.norm_source = lambda *a: (seen.append(1), inner(*a))[1]
            tr.model.float()
            data = tr.next_batch(text_batch(32, 2, 32, seed=3))
            tr.optimizer.zero_grad()
            loss, _, _ = tr.forward_loss(data)
            loss.backward()
            res[mode] = (float(loss), {n.replace("_checkpoint_wrapped_module.", ""): p.grad.clone()
                                       for n, p in tr.model.named_parameters()})
    oracle_ops.norm_source = inner
    assert len(seen) == 2 * len(tr.model.model.layers)                 # both norms of every block were announced
    for mode in ("full", "op"):
        assert res["none"][0] == pytest.approx(res[mode][0], rel=1e-6)
        for n, g in res["none"][1].items():
            torch.testing.assert_close(res[mode][1][n], g, rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------------ dataloader hook
ROOT = os.path.join(HERE, "golden", "touchdataset")


def _data_cfg(tmp_path, dirs, kind, **over):
    lst = tmp_path / "data.list"
    lst.write_text("".join(f"{d} {kind}\n" for d in dirs))
    cfg = types.SimpleNamespace(
        datapipe_type="touch_audio", datalist_path=str(lst), datalist_dev_path=str(lst), datalist_epoch=2,
        datalist_shuffling=True, datalist_sharding=False, dataset_mmap=True, dataset_shuffling=True,
        dataset_load_audio_via_segments=False, dataset_random_cut_audio=False, dataset_enable_pack=True,
        dataset_batchsize=1, dataset_text_seqlen=128, dataset_audio_seqlen=128, dataloader_drop_last_batch=False,
        dataloader_prefetch_factor=2, audio_feat_type="fbank", audiofeat_num_mel_bins=80, audiofeat_stack_length=7,
        audiofeat_stride_length=6, audiofeat_normalize=True, audiofeat_dither=0.0, audiofeat_frame_length=25,
        audiofeat_frame_shift=10, audio_speed_perturb=False, audiofeat_spec_aug=False, audiofeat_spec_sub=False,
        audiofeat_spec_trim=False, audio_min_length_in_ms_for_filter=10, audio_max_length_in_ms_for_filter=60000,
        text_min_length_in_tokens_for_filter=1, text_max_length_in_tokens_for_filter=1000, min_text_audio_ratio=0.0,
        max_text_audio_ratio=100.0)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


class _CharTok:
    bos, eos, pad = 1, 2, 0

    def tokenize(self, text, add_special_tokens=False):
        return [3 + ord(c) % 50 for c in text]


def test_build_dataloader_fn_keywords_iteration_state_and_resume(tmp_path):
    """Called with the reference's keywords; ASR shards -> device-frontend stages (oracle ops on CPU) -> packed batches;
    `state_dict()` after k batches + `load_state_dict` on a fresh loader continues with batch k+1 exactly."""
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.utils.train_spec import get_train_spec
    dirs = [os.path.join(ROOT, "synthetic", f"00000000{i}") for i in (0, 1)] + \
           [os.path.join(ROOT, "1sample_per_shard", f"00000000{i}") for i in (0, 1)]
    cfg = _data_cfg(tmp_path, dirs, "audio+metainfo")
    spec = get_train_spec("touch_audio_mi355")
    with use_ops(oracle_ops):
        make = lambda split="train": spec.build_dataloader_fn(tokenizer=_CharTok(), data_config=cfg, dp_rank=0,
                                                              dp_world_size=1, split=split)
        full = list(make())
        assert len(full) >= 3 and all(b["input_features"].shape[-1] == 80 * 7 for b in full)
        assert {"input_ids", "labels", "position_ids", "attention_mask", "sentence_lens", "num_sentence"} <= set(full[0])
        loader = make()
        it = iter(loader)
        first = [next(it) for _ in range(2)]
        state = loader.state_dict()
        loader.shutdown()
        assert set(state) == {"dp_rank_0", "world_size"}
        resumed = make()
        resumed.load_state_dict(state)
        rest = list(resumed)
        assert len(first) + len(rest) == len(full)
        for got, want in zip(first + rest, full):
            for k in ("input_ids", "labels", "attention_mask", "sentence_lens"):
                assert torch.equal(got[k], want[k]), k
            torch.testing.assert_close(got["input_features"], want["input_features"])
        assert resumed.get_epoch() == 2
        dev = list(make("dev"))                                  # evaluation split: one epoch, no shuffling
        assert 0 < len(dev) <= len(full)


_RECIPE_AUG = dict(audio_speed_perturb=True, audio_speed_perturb_speeds=[0.9, 1.0, 1.1], audio_resample_rate=16000,
                   audiofeat_spec_aug=True, audiofeat_spec_aug_num_t_mask=2, audiofeat_spec_aug_num_f_mask=2,
                   audiofeat_spec_aug_max_t=50, audiofeat_spec_aug_max_f=10, audiofeat_spec_sub=True,
                   audiofeat_spec_sub_num_t_sub=3, audiofeat_spec_sub_max_t=30, audiofeat_spec_trim=False,
                   audiofeat_spec_trim_max_t=20)          # examples/audio/sft/asr/wenetspeech/run.sh:258-270


@pytest.mark.parametrize("pack", [True, False])
def test_dataloader_with_the_asr_recipes_augmentation_chain(tmp_path, pack):
    """The reference's ASR recipe chain end to end behind `build_dataloader_fn`: resample (identity) -> speed perturbation
    -> fbank -> spec_aug -> spec_sub -> stack -> batcher (packed, or the unpacked form with `dataset_enable_pack=False`).
    The stages draw from the global `random` stream: the same seed gives the same batches, another seed different ones;
    perturbed utterances change length by 1 / speed; the augmented features differ from the plain chain's."""
    import random
    import touchnet_amd.specs  # noqa: F401
    from touchnet_amd.utils.train_spec import get_train_spec
    dirs = [os.path.join(ROOT, "synthetic", f"00000000{i}") for i in (0, 1)] +            [os.path.join(ROOT, "1sample_per_shard", f"00000000{i}") for i in (0, 1)]
    spec = get_train_spec("touch_audio_mi355")

    def run(seed, **over):
        cfg = _data_cfg(tmp_path, dirs, "audio+metainfo", datalist_epoch=1, dataset_enable_pack=pack,
                        dataset_batchsize=2, **over)
        random.seed(seed)
        with use_ops(oracle_ops):
            loader = spec.build_dataloader_fn(tokenizer=_CharTok(), data_config=cfg, dp_rank=0, dp_world_size=1, split="train")
            out = list(loader)
            loader.shutdown()
        return out
    plain = run(0)
    a, b, c = run(1, **_RECIPE_AUG), run(1, **_RECIPE_AUG), run(2, **_RECIPE_AUG)
    assert len(a) == len(b) >= 1 and sum(int(x["num_sentence"]) for x in a) == sum(int(x["num_sentence"]) for x in plain)
    for x, y in zip(a, b):
        assert torch.equal(x["input_features"], y["input_features"]) and torch.equal(x["labels"], y["labels"])
    same = len(a) == len(c) and all(x["input_features"].shape == y["input_features"].shape and
                                   torch.equal(x["input_features"], y["input_features"]) for x, y in zip(a, c))
    assert not same                                              # another seed: other speeds / stripes
    assert not (len(a) == len(plain) and all(x["input_features"].shape == y["input_features"].shape and
                                             torch.equal(x["input_features"], y["input_features"])
                                             for x, y in zip(a, plain)))
    for x in a:
        assert torch.isfinite(x["input_features"]).all()
        if not pack:
            assert x["position_ids"] is None and set(x["attention_mask"].unique().tolist()) <= {0, 1}


# ------------------------------------------------------------------------------------------------ Qwen2-Audio samples
class _QwenTok:
    """== make_golden.py::_CharTokenizer (the stand-in the reference was run with)."""
    SPECIAL = {"<|audio_bos|>": 3, "<|AUDIO|>": 4, "<|audio_eos|>": 5}
    eos_token_id, pad_token_id = 2, 0

    def convert_tokens_to_ids(self, tok):
        return self.SPECIAL[tok]

    def __call__(self, text, padding=False, return_tensors=None, add_special_tokens=True):
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
        return types.SimpleNamespace(input_ids=ids)


def test_qwen2_audio_packed_batcher_holds_the_references_samples(golden):
    """Every document of our PACKED batch == the row the reference's `dynamic_batch` builds for that sample (token ids,
    pre-shifted labels, sentence_lens, AUDIO-token count, mel features), incl. a clip longer than 30 s."""
    from touchnet_amd.models.qwen2_audio.processing_qwen2_audio import batch_qwen2_audio_packed
    g = golden("qwen2_audio_data.npz")
    rng = np.random.RandomState(int(g["wave_seed"]))
    samples = []
    for i, (d, tx) in enumerate(zip(g["durations"], g["texts"])):
        n = int(d * 16000)
        assert n == int(g[f"s{i}/n_samples"])
        samples.append({"waveform": torch.from_numpy((rng.randn(1, n) * 0.05).astype(np.float32)), "txt": str(tx),
                        "sample_rate": 16000})
    samples[2]["instruct"] = "Translate:"
    T = 4096
    cfg = types.SimpleNamespace(dataset_batchsize=2, dataset_text_seqlen=T, dataloader_drop_last_batch=False,
                                audio_max_length_in_ms_for_filter=40000, text_min_length_in_tokens_for_filter=1,
                                text_max_length_in_tokens_for_filter=100000, audiofeat_num_mel_bins=128)
    with use_ops(oracle_ops):
        batches = list(batch_qwen2_audio_packed(iter(samples), cfg, types.SimpleNamespace(tokenizer=_QwenTok())))
    si = 0
    for b in batches:
        doc = b["attention_mask"]
        clip = 0
        apos = b["audio_positions"]
        off = 0
        for r in range(doc.shape[0]):
            for d in range(1, int(doc[r].max()) + 1):
                cols = (doc[r] == d).nonzero().squeeze(1)
                ref_ids, ref_lab, ref_sl = g[f"s{si}/input_ids"], g[f"s{si}/labels"], g[f"s{si}/sentence_lens"]
                assert np.array_equal(b["input_ids"][r, cols].numpy(), ref_ids), si
                assert np.array_equal(b["labels"][r, cols].numpy(), ref_lab), si
                assert np.array_equal(b["sentence_lens"][r, cols].numpy(), ref_sl), si
                assert np.array_equal(b["position_ids"][r, cols].numpy(), np.arange(len(ref_ids)))
                n_audio = int((ref_ids == 4).sum())
                assert int(b["audio_output_lengths"][clip]) == n_audio
                want_pos = r * T + cols[torch.from_numpy(ref_ids == 4)]
                assert torch.equal(apos[off:off + n_audio], want_pos)
                frames = int(g[f"s{si}/frames"])
                mel = b["input_features"][clip, :, :frames].numpy()
                np.testing.assert_allclose(mel[:, :40], g[f"s{si}/mel_head"], atol=2e-3)
                np.testing.assert_allclose(mel[:, ::97], g[f"s{si}/mel_strided"], atol=2e-3)
                off += n_audio
                clip += 1
                si += 1
        assert off == apos.numel() and clip == b["input_features"].shape[0] == b["num_sentence"]
    assert si == len(samples)


@pytest.mark.parametrize("name", ["dyn_a", "dyn_b"])
def test_qwen2_audio_dynamic_batch_equals_the_reference(golden, name):
    """`dynamic_batch` (the reference's own unpacked batcher, processing_qwen2_audio.py:17-199): batch boundaries of the
    flush rule, padding values and every integer tensor equal the reference run on the same stream (one clip longer than
    30 s, a custom instruction, drop_last on / off); mel features from the oracle's log-mel at the reference's tolerance."""
    from touchnet_amd.models.qwen2_audio.processing_qwen2_audio import dynamic_batch
    g = golden("qwen2_audio_data.npz")
    rng = np.random.RandomState(int(g["wave_seed"]))
    samples = []
    for d, tx in zip(g["durations"], g["texts"]):
        samples.append({"waveform": torch.from_numpy((rng.randn(1, int(d * 16000)) * 0.05).astype(np.float32)),
                        "txt": str(tx), "sample_rate": 16000})
    samples[2]["instruct"] = "Translate:"
    bs, seqlen, drop = [int(v) for v in g[f"{name}/cfg"]]
    cfg = types.SimpleNamespace(dataset_batchsize=bs, dataset_text_seqlen=seqlen, dataloader_drop_last_batch=bool(drop),
                                audio_max_length_in_ms_for_filter=40000, text_min_length_in_tokens_for_filter=1,
                                text_max_length_in_tokens_for_filter=100000, audiofeat_num_mel_bins=128)
    with use_ops(oracle_ops):
        batches = list(dynamic_batch(iter(samples), cfg, types.SimpleNamespace(tokenizer=_QwenTok())))
    assert len(batches) == int(g[f"{name}/n"]) >= 2
    for j, b in enumerate(batches):
        for k in ("input_ids", "attention_mask", "labels", "shift_labels", "sentence_lens", "feature_attention_mask"):
            want = g[f"{name}/b{j}/{k}"]
            assert b[k].dtype == torch.int64 and np.array_equal(b[k].numpy(), want), (name, j, k)
        assert int(b["num_sentence"]) == int(g[f"{name}/b{j}/num_sentence"])
        assert tuple(b["input_features"].shape) == tuple(int(v) for v in g[f"{name}/b{j}/feat_shape"])
        np.testing.assert_allclose(b["input_features"].numpy()[:, ::8, ::97], g[f"{name}/b{j}/feat_strided"], atol=2e-3)
