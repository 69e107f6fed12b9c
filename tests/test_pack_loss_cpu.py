"""Twin of the reference's tests/touchnet/utils/test_pack_loss.py:133-171 for the product's loss function
(touchnet_amd/loss/cross_entropy.py behind `loss_fn`): on world 2 / 4 / 8 gloo ranks the loss of the eight sentences
evaluated BATCH-split (padded rows over the data-parallel ranks, the reference's `calc_batch_dp_loss`) equals the loss of
the same sentences PACKED into one row and split along the sequence (`calc_pack_sp_loss`: context parallelism), every rank
agrees, abs 1e-6 — with the reference's literal label vectors and the logits its run was recorded with
(tests/golden/ce_loss.npz `pack/*`, whose `pack/loss_per_sample` both must hit).  Here a rank's share is what the product's
step computes: `sentence_lens` travels with the tokens, `num_sentence` is the GLOBAL count, and the parts ADD UP
(touchnet/bin/train.py:339-343, 485-494)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

LENS = [5, 8, 3, 8, 3, 4, 6, 3]          # test_pack_loss.py:160


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, logits, labels, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle.ops as oops
        from touchnet_amd.loss.cross_entropy import cross_entropy_loss
        from touchnet_amd.models.backend import use_ops
        from touchnet_amd.utils.distributed import dist_sum
        n = len(LENS)
        sl = torch.cat([torch.full((l,), int((lab != -100).sum())) for l, lab in zip(LENS, labels.split(LENS))])
        with use_ops(oops):
            # (a) packed, sequence-split: this rank's T / world tokens of the ONE packed row
            T = logits.shape[0]
            a, b = rank * (T // world), (rank + 1) * (T // world)
            part, _ = cross_entropy_loss(logits[None, a:b], labels[None, a:b], sl[None, a:b], n)
            pack = float(dist_sum(part.reshape(1)))
            # (b) padded batch, split over dp: this rank's n / world sentences, each in its own right-padded row
            rows = list(zip(logits.split(LENS), labels.split(LENS)))[rank * (n // world):(rank + 1) * (n // world)]
            L = max(LENS)
            lg = torch.zeros(len(rows), L, logits.shape[1])
            lb = torch.full((len(rows), L), -100, dtype=torch.int64)
            ln = torch.ones(len(rows), L, dtype=torch.int64)
            for i, (x, y) in enumerate(rows):
                lg[i, :len(x)], lb[i, :len(y)] = x, y
                ln[i] = int((y != -100).sum())
            part, _ = cross_entropy_loss(lg, lb, ln, n)
            batch = float(dist_sum(part.reshape(1)))
        ret[rank] = (pack, batch)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_pack_loss(golden, world):
    g = golden("ce_loss.npz")
    logits, labels = torch.tensor(g["pack/logits"])[0], torch.tensor(g["pack/labels"])[0]
    assert logits.shape[0] == sum(LENS) and logits.shape[0] % world == 0 and len(LENS) % world == 0
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), logits, labels, ret), nprocs=world, join=True)
        results = dict(ret)
    packs, batches = [results[r][0] for r in range(world)], [results[r][1] for r in range(world)]
    assert len(set(packs)) == 1 and len(set(batches)) == 1, (packs, batches)      # every rank agrees
    assert packs[0] == pytest.approx(batches[0], abs=1e-6)
    assert packs[0] == pytest.approx(float(g["pack/loss_per_sample"]), abs=1e-6)   # = what the reference's loss gave
