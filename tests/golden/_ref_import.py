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
