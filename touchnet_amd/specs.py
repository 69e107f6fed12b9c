"""TrainSpec registrations of the MI355X path — the counterpart of touchnet/__init__.py:35-117.

Model families are selected by `training_model_name`, like in the reference:
    llama_mi355         <- "llama"        (packed text pre-training; LlamaForCausalLM / Qwen2ForCausalLM shapes)
    touch_audio_mi355   <- "touch_audio"  (LlamaForASR: projector + packed ASR pairs)
    qwen2_audio_mi355   <- "qwen2_audio"  (Qwen2-Audio-7B, packed variant)
INTEGRATION.md shows the three-line shim that puts these into TouchNet's own registry.
"""
from touchnet_amd.loss.cross_entropy import cross_entropy_loss
from touchnet_amd.models import llama, qwen2_audio, touch_audio
from touchnet_amd.models.helper_func import apply_fsdp
from touchnet_amd.utils.metrics import accuracy
from touchnet_amd.utils.optimizer import FusedAdamW, linear_warmup_linear_decay
from touchnet_amd.utils.train_spec import TrainSpec, _train_specs, get_train_spec, register_train_spec  # noqa: F401


def _parallelize(model, dp_mesh, job):
    import torch
    return apply_fsdp(model, dp_mesh,
                      param_dtype=getattr(torch, getattr(job, "training_mixed_precision_param", "bfloat16")),
                      reduce_dtype=getattr(torch, getattr(job, "training_mixed_precision_reduce", "float32")),
                      reshard_after_forward_policy=getattr(job, "training_fsdp_reshard_after_forward", "never"))


def _build_optimizers(model_parts, job):
    params = [p for m in model_parts for p in m.parameters()]
    return FusedAdamW(params, lr=job.lr_scheduler_lr, weight_decay=job.optimizer_weight_decay,
                      max_norm=job.training_max_norm)


def _build_lr(optimizers, job):
    return lambda step: job.lr_scheduler_lr * linear_warmup_linear_decay(
        step, job.lr_scheduler_warmup_steps, job.lr_scheduler_steps)


def _synthetic_loader(**kw):
    raise NotImplementedError("the MI355X path is fed by touchnet's own dataloader over touchnet_amd.data.datapipe "
                              "(INTEGRATION.md) or by touchnet_amd.data.synthetic in benchmarks")


def _build_tokenizer(args, **kwargs):
    """touchnet/tokenizer/tokenizer.py:321-334: the BEST-RQ label tokenizer runs on the device; text tokenizers are
    out of scope and stay the reference's (this raises NotImplementedError for them)."""
    from touchnet_amd.tokenizer import build_tokenizer
    return build_tokenizer(args, **kwargs)


def _spec(name, mod, model_cls, config_cls):
    return TrainSpec(name=name, model_cls=model_cls, config_cls=config_cls, parallelize_fn=_parallelize,
                     pipelining_fn=None, build_optimizers_fn=_build_optimizers, build_lr_schedulers_fn=_build_lr,
                     build_dataloader_fn=_synthetic_loader, build_tokenizer_fn=_build_tokenizer, loss_fn=cross_entropy_loss,
                     acc_fn=accuracy, additional_pre_init_fn=mod.pre_init, additional_post_init_fn=mod.post_init,
                     get_num_flop_per_token_fn=mod.get_num_flop_per_token, get_num_params_fn=mod.get_num_params)


def register_all():
    for spec in (
        _spec("llama_mi355", llama, llama.PackedCausalLM, llama.DecoderConfig),
        _spec("touch_audio_mi355", touch_audio, touch_audio.TouchAudioForCausalLM, touch_audio.TouchAudioConfig),
        _spec("qwen2_audio_mi355", qwen2_audio, qwen2_audio.Qwen2AudioPackedForConditionalGeneration,
              qwen2_audio.Qwen2AudioConfig),
    ):
        if spec.name not in _train_specs:
            register_train_spec(spec)


register_all()
