"""Model-family plugin registry with the surface TouchNet's trainer consumes.

Surface parity (field names AND order, argument meaning, error behaviour) with touchnet/utils/train_spec.py:25-68 —
`tests/test_boundary.py` holds the reference's field list and the call signatures of its `train.py` call sites as a
fixture generated from the reference (tests/golden/make_golden.py) and checks this module and the registered specs
against it:
  * `TrainSpec` — one record per `training_model_name`; the sixteen slots are the hooks touchnet/bin/train.py calls;
    all positional, none defaulted, in the reference's order (a spec built positionally for one registry fits the other)
  * `register_train_spec(spec)` — ValueError when the name is taken
  * `get_train_spec(name)`      — ValueError when the name is unknown
  * `apply_to_train_specs(fn)`  — rewrite every registered spec through `fn`
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict


@dataclass
class TrainSpec:
    name: str
    model_cls: Any                        # nn.Module subclass, constructible on the meta device (train.py:179-182)
    config_cls: Any                       # `.from_json_file(path)` (train.py:126-127)
    parallelize_fn: Callable              # (model, world_mesh, parallel_dims, job_config) -> model   train.py:259-261
    pipelining_fn: Callable               # None here: pipeline parallelism is out of scope
    build_optimizers_fn: Callable         # (model_parts, job_config)                                  train.py:298
    build_lr_schedulers_fn: Callable      # (optimizers, job_config)                                   train.py:299
    build_dataloader_fn: Callable         # (tokenizer=, data_config=, dp_rank=, dp_world_size=, split=)  :157-170
    build_tokenizer_fn: Callable          # (tokenizer_config, **special_tokens)                       train.py:150-156
    loss_fn: Callable                     # (pred, labels, sentence_lens, num_sentence) -> (per_sample, per_token)
    acc_fn: Callable                      # (pred, labels) -> 0-d tensor, or None to skip              train.py:449-452
    additional_pre_init_fn: Callable      # (job_config)                                               train.py:121-122
    additional_post_init_fn: Callable     # (model, init_device)                                       train.py:274-281
    get_num_flop_per_token_fn: Callable   # (num_params, model_config, seq_len) -> int
    get_num_params_fn: Callable           # (model, exclude_embedding=False) -> int
    build_metrics_processor_fn: Callable  # (job_config, parallel_dims) — the reference's own is used (logging: out of scope)


_train_specs: Dict[str, TrainSpec] = {}


def register_train_spec(train_spec: TrainSpec) -> None:
    if train_spec.name in _train_specs:
        raise ValueError(f"Model {train_spec.name} is already registered.")
    _train_specs[train_spec.name] = train_spec


def get_train_spec(name: str) -> TrainSpec:
    if name not in _train_specs:
        raise ValueError(f"Model {name} is not registered.")
    return _train_specs[name]


def apply_to_train_specs(func: Callable[[TrainSpec], TrainSpec]) -> None:
    for name in list(_train_specs):
        _train_specs[name] = func(_train_specs[name])
