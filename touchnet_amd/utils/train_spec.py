"""Model-family plugin registry with the surface TouchNet's trainer consumes.

Surface parity (names, argument meaning, error behaviour) with touchnet/utils/train_spec.py:25-68:
  * `TrainSpec` — one record per `training_model_name`; the sixteen slots are the hooks
    touchnet/bin/train.py calls (model / config classes, parallelize, optimizer + scheduler builders,
    dataloader / tokenizer builders, loss / acc functions, pre/post-init hooks, flop + parameter counters,
    metrics processor builder).
  * `register_train_spec(spec)` — ValueError when the name is taken.
  * `get_train_spec(name)`      — ValueError when the name is unknown.
  * `apply_to_train_specs(fn)`  — rewrite every registered spec through `fn`.
INTEGRATION.md shows how the MI355X specs are entered into TouchNet's own registry.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Callable, Dict, Optional

_SLOTS = (
    # (field name, required)
    ("name", True), ("model_cls", True), ("config_cls", True), ("parallelize_fn", True),
    ("pipelining_fn", False), ("build_optimizers_fn", True), ("build_lr_schedulers_fn", True),
    ("build_dataloader_fn", True), ("build_tokenizer_fn", False), ("loss_fn", True), ("acc_fn", False),
    ("additional_pre_init_fn", False), ("additional_post_init_fn", False),
    ("get_num_flop_per_token_fn", True), ("get_num_params_fn", True), ("build_metrics_processor_fn", False),
)

TrainSpec = dataclasses.make_dataclass(
    "TrainSpec",
    [(n, Any) if req else (n, Optional[Callable], dataclasses.field(default=None)) for n, req in
     sorted(_SLOTS, key=lambda t: not t[1])],
)
TrainSpec.__doc__ = "Plugin record for one model family (fields = the hooks of touchnet/bin/train.py)."


class _Registry:
    def __init__(self):
        self._by_name: Dict[str, Any] = {}

    def add(self, spec) -> None:
        if spec.name in self._by_name:
            raise ValueError(f"Model {spec.name} is already registered.")
        self._by_name[spec.name] = spec

    def get(self, name: str):
        try:
            return self._by_name[name]
        except KeyError:
            raise ValueError(f"Model {name} is not registered.") from None

    def map_inplace(self, fn: Callable) -> None:
        self._by_name = {k: fn(v) for k, v in self._by_name.items()}

    def __contains__(self, name: str) -> bool:
        return name in self._by_name


_REGISTRY = _Registry()
_train_specs = _REGISTRY          # `name in _train_specs` keeps working for callers of the old module dict


def register_train_spec(train_spec) -> None:
    _REGISTRY.add(train_spec)


def get_train_spec(name: str):
    return _REGISTRY.get(name)


def apply_to_train_specs(func: Callable) -> None:
    _REGISTRY.map_inplace(func)
