"""TrainSpec registry with the reference's exact surface (touchnet/utils/train_spec.py:25-68):
same dataclass fields, `register_train_spec` (ValueError on duplicates), `get_train_spec`,
`apply_to_train_specs`.  INTEGRATION.md shows how these specs register into TouchNet's own registry so
that touchnet/bin/train.py drives the MI355X path unchanged."""
from dataclasses import dataclass
from typing import Any, Callable, Optional, Type

import torch.nn as nn


@dataclass
class TrainSpec:
    name: str
    model_cls: Type[nn.Module]
    config_cls: Any
    parallelize_fn: Callable
    pipelining_fn: Optional[Callable]
    build_optimizers_fn: Callable
    build_lr_schedulers_fn: Callable
    build_dataloader_fn: Callable
    build_tokenizer_fn: Optional[Callable]
    loss_fn: Callable
    acc_fn: Optional[Callable]
    additional_pre_init_fn: Optional[Callable]
    additional_post_init_fn: Optional[Callable]
    get_num_flop_per_token_fn: Callable
    get_num_params_fn: Callable
    build_metrics_processor_fn: Optional[Callable] = None


_train_specs = {}


def register_train_spec(train_spec: TrainSpec) -> None:
    if train_spec.name in _train_specs:
        raise ValueError(f"Model {train_spec.name} is already registered.")
    _train_specs[train_spec.name] = train_spec


def get_train_spec(name: str) -> TrainSpec:
    if name not in _train_specs:
        raise ValueError(f"Model {name} is not registered.")
    return _train_specs[name]


def apply_to_train_specs(func: Callable[[TrainSpec], TrainSpec]) -> None:
    for name, spec in list(_train_specs.items()):
        _train_specs[name] = func(spec)
