"""TrainSpec registrations of the MI355X path — the counterpart of touchnet/__init__.py:35-117.

Model families are selected by `training_model_name`, like in the reference:
    llama_mi355         <- "llama"        (packed text pre-training; LlamaForCausalLM / Qwen2ForCausalLM shapes)
    touch_audio_mi355   <- "touch_audio"  (LlamaForASR: projector + packed ASR pairs)
    qwen2_audio_mi355   <- "qwen2_audio"  (Qwen2-Audio-7B, packed variant)
    kimi_audio_mi355    <- "kimi_audio"   (Kimi-Audio-7B decoder: Qwen2 stack + mimo branch, text-logit CE; config E groundwork)
INTEGRATION.md shows the three-line shim that puts these into TouchNet's own registry.
"""
from touchnet_amd.data.dataloader import build_dataloader
from touchnet_amd.loss.cross_entropy import cross_entropy_loss
from touchnet_amd.models import kimi_audio, llama, qwen2_audio, touch_audio
from touchnet_amd.models.parallelize import parallelize_packed
from touchnet_amd.utils.metrics import MI355X_BF16_DENSE_PEAK, accuracy
from touchnet_amd.utils.optimizer import FusedAdamW, LRScheduler
from touchnet_amd.utils.train_spec import TrainSpec, _train_specs, get_train_spec, register_train_spec  # noqa: F401


# tests drive the host logic on CPU with a torch optimizer: `(named_params_or_shards, process_group) -> optimizer`
OPTIMIZER_FACTORY = None


def _build_optimizers(model_parts, job):
    from touchnet_amd.models.tensor_parallel import tp_param_ids
    if len(model_parts) == 1 and getattr(model_parts[0], "_tn_flat_dp", None) is not None:
        # `parallelize_fn` chose the flat data-parallel engine (training_dp_engine=flat): the engine is built HERE, on the
        # parameters the trainer has materialised in between, and what comes back drives it through the trainer's own
        # zero_grad() / step() calls (utils/zero_dp.py, bottom)
        from touchnet_amd.utils.zero_dp import build_flat_engine_optimizer
        make = OPTIMIZER_FACTORY or (lambda shards, group: FusedAdamW(
            shards, lr=job.lr_scheduler_lr, weight_decay=job.optimizer_weight_decay, max_norm=job.training_max_norm,
            process_group=group))
        return build_flat_engine_optimizer(model_parts[0], make)
    many = len(model_parts) > 1
    params = [(f"{i}.{n}" if many else n, p) for i, m in enumerate(model_parts) for n, p in m.named_parameters()]
    tp_group, tp_ids = tp_param_ids(model_parts)
    return FusedAdamW(params, lr=job.lr_scheduler_lr, weight_decay=job.optimizer_weight_decay,
                      max_norm=job.training_max_norm, tp_group=tp_group, tp_param_ids=tp_ids)


def _build_lr(optimizers, job):
    """`build_lr_schedulers_fn(optimizers, job_config)` (touchnet/bin/train.py:299): an object with step() /
    state_dict() / load_state_dict() like the reference's LRSchedulersContainer."""
    return LRScheduler(optimizers, job.lr_scheduler_lr, job.lr_scheduler_warmup_steps, job.lr_scheduler_steps,
                       getattr(job, "lr_scheduler_lr_min", 0.0))


def _build_dataloader(tokenizer=None, data_config=None, dp_rank=0, dp_world_size=1, split="train"):
    """Called with exactly these keywords at touchnet/bin/train.py:157-170."""
    return build_dataloader(data_config, tokenizer, dp_rank, dp_world_size, split)


def _build_metrics_processor(job_config, parallel_dims):
    """touchnet/bin/train.py:185.  Logging / MFU bookkeeping is the reference's own (out of scope here); its peak-FLOPS
    table has no MI-series entry (utils/metrics.py:67-100 falls back to A100), so the MI355X dense bf16 peak is set."""
    from touchnet.utils.metrics import build_metrics_processor
    mp = build_metrics_processor(job_config, parallel_dims)
    mp.gpu_peak_flops = MI355X_BF16_DENSE_PEAK
    return mp


def _build_tokenizer(args, **kwargs):
    """touchnet/tokenizer/tokenizer.py:321-334: the BEST-RQ label tokenizer runs on the device; text tokenizers are
    out of scope and stay the reference's (this raises NotImplementedError for them)."""
    from touchnet_amd.tokenizer import build_tokenizer
    return build_tokenizer(args, **kwargs)


def _spec(name, mod, model_cls, config_cls):
    return TrainSpec(name=name, model_cls=model_cls, config_cls=config_cls, parallelize_fn=parallelize_packed,
                     pipelining_fn=None, build_optimizers_fn=_build_optimizers, build_lr_schedulers_fn=_build_lr,
                     build_dataloader_fn=_build_dataloader, build_tokenizer_fn=_build_tokenizer,
                     loss_fn=cross_entropy_loss, acc_fn=accuracy, additional_pre_init_fn=mod.pre_init,
                     additional_post_init_fn=mod.post_init, get_num_flop_per_token_fn=mod.get_num_flop_per_token,
                     get_num_params_fn=mod.get_num_params, build_metrics_processor_fn=_build_metrics_processor)


def register_all():
    for spec in (
        _spec("llama_mi355", llama, llama.PackedCausalLM, llama.DecoderConfig),
        _spec("touch_audio_mi355", touch_audio, touch_audio.TouchAudioForCausalLM, touch_audio.TouchAudioConfig),
        _spec("qwen2_audio_mi355", qwen2_audio, qwen2_audio.Qwen2AudioPackedForConditionalGeneration,
              qwen2_audio.Qwen2AudioConfig),
        _spec("kimi_audio_mi355", kimi_audio, kimi_audio.KimiAudioPackedForCausalLM, kimi_audio.KimiAudioConfig),
    ):
        if spec.name not in _train_specs:
            register_train_spec(spec)


register_all()
