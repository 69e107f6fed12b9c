"""Which op namespace the model modules call.

Default (and the only thing the product ever selects): ``touchnet_amd.functional`` = the HIP kernels.
``use_ops`` exists so that the CPU test-suite can drive the *host logic* (module wiring, packers,
sharding, collectives under gloo) with the oracle's op set; nothing in this package calls it.
"""
from contextlib import contextmanager

_OPS = None


def ops():
    global _OPS
    if _OPS is None:
        from touchnet_amd import functional
        _OPS = functional
    return _OPS


@contextmanager
def use_ops(namespace):
    global _OPS
    prev, _OPS = _OPS, namespace
    try:
        yield
    finally:
        _OPS = prev
