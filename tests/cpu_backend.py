"""CPU stand-in for the two HIP entry points, built on the ORACLE — for host-logic tests only.

`-m "not gpu"` tests exercise the product's Python (blocks, UNet, sampler, sharding) on a machine
without a GPU by swapping sta.ops.pack_kv / sta.ops.xattn_blend for the oracle's fused form. The
product itself has no such switch: outside these tests a missing GPU/library raises.
"""
import contextlib

import torch

from oracle import xattn_oracle as orc


class CpuPacked:
    def __init__(self, k, v, heads, n_img):
        self.k, self.v, self.heads, self.n_img = k, v, heads, n_img
        self.n_ctx, self.M, self.C, self.dtype = k.shape[0] // n_img, k.shape[1], k.shape[2], k.dtype


def _pack_kv(k, v, heads, out=None, n_img=1):
    return CpuPacked(k.detach().clone(), v.detach().clone(), heads, n_img)


def _xattn_blend(q, coef, packed, mask, scale):
    K, I, n = packed.n_ctx - 2, packed.n_img, packed.n_ctx
    outs = []
    for i in range(I):
        mi = mask.reshape(I, -1)[i] if K else None
        m = torch.stack([(mi >> j) & 1 for j in range(K)]).bool() if K else torch.zeros((0, q.shape[1]), dtype=torch.bool)
        c = coef.reshape(I, K)[i] if K else torch.zeros(0, dtype=q.dtype)
        outs.append(orc.fused_xattn(q[2 * i:2 * i + 2], packed.k[n * i:n * (i + 1)], packed.v[n * i:n * (i + 1)], m,
                                    c.to(q.dtype), packed.heads, scale))
    return torch.cat(outs)


@contextlib.contextmanager
def oracle_ops():
    from sta import ops
    saved = ops.pack_kv, ops.xattn_blend
    ops.pack_kv, ops.xattn_blend = _pack_kv, _xattn_blend
    try:
        yield
    finally:
        ops.pack_kv, ops.xattn_blend = saved
