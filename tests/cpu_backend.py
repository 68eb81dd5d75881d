"""CPU stand-in for the two HIP entry points, built on the ORACLE — for host-logic tests only.

`-m "not gpu"` tests exercise the product's Python (blocks, UNet, sampler, sharding) on a machine
without a GPU by swapping sta.ops.pack_kv / sta.ops.xattn_blend for the oracle's fused form. The
product itself has no such switch: outside these tests a missing GPU/library raises.
"""
import contextlib

import torch

from oracle import xattn_oracle as orc


class CpuPacked:
    def __init__(self, k, v, heads):
        self.k, self.v, self.heads = k, v, heads
        self.n_ctx, self.M, self.C, self.dtype = k.shape[0], k.shape[1], k.shape[2], k.dtype


def _pack_kv(k, v, heads, out=None):
    return CpuPacked(k.detach().clone(), v.detach().clone(), heads)


def _xattn_blend(q, coef, packed, mask, scale):
    K = packed.n_ctx - 2
    m = torch.stack([(mask >> i) & 1 for i in range(K)]).bool() if K else torch.zeros((0, q.shape[1]), dtype=torch.bool)
    c = coef if K else torch.zeros(0, dtype=q.dtype)
    return orc.fused_xattn(q, packed.k, packed.v, m, c.to(q.dtype), packed.heads, scale)


@contextlib.contextmanager
def oracle_ops():
    from sta import ops
    saved = ops.pack_kv, ops.xattn_blend
    ops.pack_kv, ops.xattn_blend = _pack_kv, _xattn_blend
    try:
        yield
    finally:
        ops.pack_kv, ops.xattn_blend = saved
