"""SURVEY.md §8b hook 1 (module swap, the liger precedent): `apply_mi355_kernels_to_llama()` /
`apply_mi355_kernels_to_qwen2()` are what a TouchNet `additional_pre_init_fn` calls in place of
`apply_liger_kernel_to_llama()` (touchnet/models/llama/__init__.py:11-15, qwen2_audio/__init__.py:252-260) to keep
transformers' own model classes and run them on the MI355X kernels:

    RMSNorm.forward        -> tn_rmsnorm_fwd/bwd               (functional.rms_norm)
    MLP.forward            -> one autograd node: two GEMMs, SwiGLU with transposed outputs, down GEMM, forward-layout
                              weight-/input-gradient GEMMs   (functional.swiglu_mlp)
    attention              -> AttentionInterface "mi355_packed" (integrations/hf_attention.py), selected by
                              `"attn_implementation": "mi355_packed"` in the model JSON

RoPE stays HF's: its `apply_rotary_pos_emb` works on [B, Nh, T, D] (heads-major) tensors, the HIP kernel on
[B, T, Nh, D]; swapping it would add two transposes per layer.  (Our own `PackedCausalLM` keeps everything in the
[B, T, Nh, D] layout and fuses the residual adds into the norms as well; the patch is the minimal-intrusion route.)
The patches are class-level and idempotent; `undo()` restores the originals (tests).
"""
from __future__ import annotations

import importlib

from touchnet_amd.models.backend import ops

_saved = {}


def _norm_forward(self, hidden_states):
    return ops().rms_norm(hidden_states, self.weight, self.variance_epsilon)


def _mlp_forward(self, x):
    if (getattr(self.gate_proj, "bias", None) is not None or getattr(self.up_proj, "bias", None) is not None
            or getattr(self.down_proj, "bias", None) is not None):
        raise NotImplementedError("mi355 MLP patch: biased MLP projections are not supported")
    return ops().swiglu_mlp(x, self.gate_proj.weight, self.up_proj.weight, self.down_proj.weight)


def _apply(module_name: str, prefix: str, rms_norm: bool, mlp: bool, attention: bool):
    mod = importlib.import_module(module_name)
    if rms_norm:
        cls = getattr(mod, f"{prefix}RMSNorm")
        _saved.setdefault((cls, "forward"), cls.forward)
        cls.forward = _norm_forward
    if mlp:
        cls = getattr(mod, f"{prefix}MLP")
        _saved.setdefault((cls, "forward"), cls.forward)
        cls.forward = _mlp_forward
    if attention:
        from . import hf_attention
        hf_attention.register()


def apply_mi355_kernels_to_llama(rms_norm: bool = True, mlp: bool = True, attention: bool = True) -> None:
    _apply("transformers.models.llama.modeling_llama", "Llama", rms_norm, mlp, attention)


def apply_mi355_kernels_to_qwen2(rms_norm: bool = True, mlp: bool = True, attention: bool = True) -> None:
    _apply("transformers.models.qwen2.modeling_qwen2", "Qwen2", rms_norm, mlp, attention)


def undo() -> None:
    for (cls, name), fn in _saved.items():
        setattr(cls, name, fn)
    _saved.clear()
