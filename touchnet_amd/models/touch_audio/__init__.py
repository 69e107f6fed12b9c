"""TouchAudio (LlamaForASR) adapter (mirrors touchnet/models/touch_audio/__init__.py:15-74)."""
from ..llama import get_num_flop_per_token as _llama_flops
from ..llama import get_num_params, post_init, pre_init  # noqa: F401
from .modeling_touch_audio import TouchAudioConfig, TouchAudioForCausalLM  # noqa: F401
from .processing_touch_audio import batch_audio_packed, batch_pairaudio_pairtext_packed  # noqa: F401


def get_num_flop_per_token(num_params, model_config, seq_len):
    return _llama_flops(num_params, model_config, seq_len)
