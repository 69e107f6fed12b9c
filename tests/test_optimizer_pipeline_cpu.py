"""Host logic of the pipelined optimizer update (utils/optimizer.py: update_groups, pipeline_updates_under_forward): which
parameters form a group, which hook waits for which group, and the launch order learnt from the first forward.  The
device side (side stream, events, bit-identical training) is tests/test_models_gpu.py."""
import torch
from torch import nn

from touchnet_amd.utils.optimizer import pipeline_updates_under_forward, update_groups


class _Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.a, self.b = nn.Linear(4, 4), nn.Linear(4, 4)

    def forward(self, x):
        return self.b(self.a(x))


class _Silent(nn.Module):
    """a repeated block whose PARENT reads the weights (never entered through __call__)"""

    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.ones(4))


class _Model(nn.Module):
    def __init__(self):
        super().__init__()
        self.embed = nn.Embedding(8, 4)
        self.decoder = nn.ModuleList([_Block() for _ in range(3)])
        self.mid = nn.LayerNorm(4)
        self.tower = nn.ModuleList([_Block() for _ in range(2)])      # registered LAST, used FIRST
        self.scales = nn.ModuleList([_Silent() for _ in range(2)])
        self.head = nn.Linear(4, 8)

    def forward(self, ids):
        x = self.embed(ids)
        for t in self.tower:
            x = t(x)
        for s in self.scales:
            x = x * s.w
        for d in self.decoder:
            x = d(x)
        return self.head(self.mid(x))


class _Recorder:
    """stands in for FusedAdamW: records what the hooks ask for"""

    def __init__(self, model):
        self.names = [n for n, _ in model.named_parameters()]
        self.log, self.order, self.groups = [], None, None

    def pipeline_updates(self, groups):
        self.groups = groups

    def set_launch_order(self, order):
        self.order = list(order)

    def wait_group(self, gi):
        self.log.append(gi)

    def wait_updates(self):
        self.log.append("all")


def test_groups_partition_the_parameters_in_registration_order():
    m = _Model()
    names = [n for n, _ in m.named_parameters()]
    groups, blocks = update_groups(m, names)
    assert sorted(i for g in groups for i in g) == list(range(len(names)))
    assert [names[i] for i in groups[0]] == ["embed.weight", "mid.weight", "mid.bias"]            # outside the blocks
    assert [names[i] for i in groups[-1]] == ["head.weight", "head.bias"]                          # behind the last block
    assert len(blocks) == 3 + 2 + 2 and [gi for _, gi in blocks] == list(range(1, 8))
    assert [names[i] for i in groups[1]] == [f"decoder.0.{l}.{w}" for l in "ab" for w in ("weight", "bias")]
    # activation-checkpoint wrappers keep the optimizer's (stripped) names
    from torch.distributed.algorithms._checkpoint.checkpoint_wrapper import checkpoint_wrapper
    for i in range(3):
        m.decoder[i] = checkpoint_wrapper(m.decoder[i])
    g2, b2 = update_groups(m, names)
    assert g2 == groups and [gi for _, gi in b2] == [gi for _, gi in blocks]


def test_hooks_wait_per_block_and_learn_the_order_of_use():
    m = _Model()
    opt = _Recorder(m)
    pipeline_updates_under_forward(m, opt)
    late = len(opt.groups) - 1
    ids = torch.arange(6).reshape(2, 3) % 8
    m(ids)
    # the learning forward waits for everything at the entry, then per block in the order of USE: tower (groups 4, 5),
    # decoder (1, 2, 3); the `scales` blocks (6, 7) are never entered
    assert opt.log[:2] == ["all", 0] and [g for g in opt.log[2:] if g != late] == [4, 5, 1, 2, 3]
    # launches follow the use: outside-the-blocks first, then the silent blocks (waited for at the entry from now on) ...
    assert opt.order[:3] == [0, 6, 7] and opt.order[3:] == [4, 5, 1, 2, 3, late] and sorted(opt.order) == list(range(late + 1))
    opt.log.clear()
    m(ids)
    assert opt.log[:4] == [0, late, 6, 7] or opt.log[:3] == [0, 6, 7]
    assert "all" not in opt.log
    rest = [g for g in opt.log if g not in (0, 6, 7, late)]
    assert rest == [4, 5, 1, 2, 3]
    assert late in opt.log                         # (the last REGISTERED block with a hook is `scales.1`: silent -> entry)
    m.state_dict()
    assert opt.log[-1] == "all"
