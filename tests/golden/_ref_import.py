"""Import recipe for the (read-only) reference at /root/reference.

Only used by tests/golden/make_golden.py, which runs in the build container to
produce the committed fixtures.  Nothing here travels to the GPU box in a usable
form: /root/reference does not exist there and no test imports this module.
"""
import importlib.util
import sys
import types
from unittest.mock import MagicMock

REF = "/root/reference"


def install():
    sys.dont_write_bytecode = True
    import transformers  # noqa: F401  (must be first: its lazy loader probes torchaudio)

    def bare(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    bare("touchnet", f"{REF}/touchnet")
    bare("touchnet.models", f"{REF}/touchnet/models")
    for sub in ("llama", "touch_audio", "qwen2_audio", "kimi_audio"):
        bare(f"touchnet.models.{sub}", f"{REF}/touchnet/models/{sub}")
    for x in ("librosa", "torchaudio", "torchaudio.compliance", "torchaudio.compliance.kaldi",
              "liger_kernel", "liger_kernel.transformers", "torchdata",
              "torchdata.stateful_dataloader"):
        if x not in sys.modules:
            sys.modules[x] = MagicMock()


def load_file_as(name, relpath):
    spec = importlib.util.spec_from_file_location(name, f"{REF}/{relpath}")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_kimi_modeling():
    """touchnet/models/kimi_audio/modeling_kimi_audio.py under the INSTALLED transformers 5.x (the reference pins 4.51.3,
    which is not in this image).  Three names it imports from `transformers.models.qwen2.modeling_qwen2` no longer exist:
    two docstring constants (only ever passed to docstring decorators) and a TypedDict used in a type annotation.  They
    are put back — in this process's copy of the third-party module, nowhere on disk — as an empty string / an empty
    TypedDict, which cannot influence any computed value.  Build container only, like everything in this file."""
    from typing import TypedDict

    import transformers.models.qwen2.modeling_qwen2 as mq

    class KwargsForCausalLM(TypedDict, total=False):
        pass
    for name, value in (("QWEN2_INPUTS_DOCSTRING", ""), ("QWEN2_START_DOCSTRING", ""),
                        ("KwargsForCausalLM", KwargsForCausalLM)):
        if not hasattr(mq, name):
            setattr(mq, name, value)
    import importlib
    return importlib.import_module("touchnet.models.kimi_audio.modeling_kimi_audio")


def adapt_decoder_layers_to_4_51(layers):
    """The second drift between 4.51.3 and 5.x that the reference's MoonshotKimiaModel.forward meets: it calls
    `decoder_layer(..., past_key_value=...)` and reads `layer_outputs[0]`; a 5.x Qwen2DecoderLayer names the argument
    `past_key_values` and returns the hidden-state TENSOR (so `[0]` would silently drop the batch dimension).  Each layer's
    forward is wrapped to take the old keyword and return the old 1-tuple; the layer's arithmetic is untouched."""
    for layer in layers:
        inner = layer.forward

        def forward(hidden_states, *args, _inner=inner, past_key_value=None, output_attentions=False, cache_position=None,
                    **kw):
            return (_inner(hidden_states, *args, past_key_values=past_key_value, **kw),)
        layer.forward = forward
