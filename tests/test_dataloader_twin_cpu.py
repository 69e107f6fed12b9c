"""Twin of the reference's tests/touchnet/data/test_dataloader.py:37-118 for the product's LowLevelTouchDatapipe: fake
`texttoken` shards (one per rank of nnodes x nproc_per_node, `max_epoch` one-token documents each, written with the
product's DataBuilder twin, touchnet_amd/data/builder.py), sharded over dp ranks, then over dataloader workers, with a mid-epoch resume.
  * num_workers = 0: the datapipe itself, broken at `break_point` and resumed from its own state_dict — the state the
    loader persists; as in the reference, the sample in flight at the break is yielded again (the counter moves behind the
    yield).
  * num_workers > 0 (no break point — the reference has none for > 1 worker either): torch's DataLoader forks the workers,
    the datapipe shards its lists by `get_worker_info()`, batches come round-robin: the reference's interleaving formula.
    (Its three one-worker resume cases snapshot worker state through torchdata's StatefulDataLoader, which is not in
    this image; the product's own loader runs the datapipe on a thread of the trainer process, tests/test_boundary.py.)"""
import os
import types

import numpy as np
import pytest
import torch

from touchnet_amd.data.datapipe import LowLevelTouchDatapipe


def _write_texttoken_shard(prefix, docs):
    """build_fake_data of the reference's test (:13-33) with the product's DataBuilder: one-sentence documents"""
    from touchnet_amd.data.builder import DataBuilder
    os.makedirs(prefix, exist_ok=True)
    b = DataBuilder(f"{prefix}/texttoken.bin", np.uint16)
    for d in docs:
        b.add_item(torch.IntTensor(d))
        b.end_document()
    b.finalize(f"{prefix}/texttoken.idx")


def _fake_data(root, nnodes, nproc_per_node, max_epoch):
    shards = []
    for i in range(nnodes * nproc_per_node):
        prefix = f"{root}/shards_{i}"
        _write_texttoken_shard(prefix, [[i * max_epoch + j] for j in range(max_epoch)])
        shards.append(prefix)
    with open(f"{root}/data.list", "w", encoding="utf8") as f:
        for name in shards:
            f.write(f"{name} texttoken\n")
    return f"{root}/data.list"


@pytest.mark.parametrize("nnodes,nproc_per_node,max_epoch,num_workers,dp_rank,dp_worldsize,break_point", [
    (4, 8, 6, 0, 3, 8, 12), (4, 8, 6, 0, 3, 8, 15), (4, 8, 6, 0, 3, 8, 5), (4, 8, 6, 0, 3, 8, 24),
    (4, 8, 6, 1, 3, 8, -1), (1, 8, 6, 4, 1, 2, -1), (1, 8, 6, 2, 1, 4, -1), (4, 8, 6, 4, 3, 8, -1), (2, 8, 6, 4, 0, 2, -1),
])
def test_dataloader(tmp_path, nnodes, nproc_per_node, max_epoch, num_workers, dp_rank, dp_worldsize, break_point):
    if num_workers > 0:
        assert (nnodes * nproc_per_node) % (dp_worldsize * num_workers) == 0
    total = nnodes * nproc_per_node * max_epoch
    config = types.SimpleNamespace(datalist_path=_fake_data(str(tmp_path), nnodes, nproc_per_node, max_epoch),
                                   datalist_sharding=True, datalist_shuffling=False, dataset_shuffling=False,
                                   dataset_mmap=True, datalist_epoch=1)
    loaded, state = [], {}
    pipe = LowLevelTouchDatapipe(config, dp_rank, dp_worldsize)
    loader = pipe if num_workers == 0 else torch.utils.data.DataLoader(pipe, batch_size=None, num_workers=num_workers,
                                                                         prefetch_factor=4)
    for i, data in enumerate(loader):
        if i == break_point:
            state = pipe.state_dict()
            break
        assert len(data["input_ids"]) == 1
        loaded.append(int(data["input_ids"][0]))
    if state:                                                     # resume from the mid-epoch state
        pipe = LowLevelTouchDatapipe(config, dp_rank, dp_worldsize)
        pipe.load_state_dict(state)
        loaded += [int(d["input_ids"][0]) for d in pipe]
    expected = np.arange(total, dtype=np.int32).reshape(-1, max_epoch)[dp_rank::dp_worldsize, :]
    if num_workers > 0:                                           # (the reference's interleaving of the workers' streams)
        parts = [expected[i::num_workers, :].reshape(1, -1) for i in range(num_workers)]
        expected = np.concatenate([p for p in parts if p.shape[-1] > 0], axis=0).transpose()
    assert np.array_equal(np.array(loaded, dtype=np.int32), expected.reshape(-1))
