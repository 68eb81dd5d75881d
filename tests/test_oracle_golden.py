"""The oracle (oracle/xattn_oracle.py) pinned against golden vectors produced by the REFERENCE's own
modules (oracle/gen_golden.py). CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import golden_inputs as gi
from oracle import xattn_oracle as orc
from sta.synth import seeded_fill_

G = gi.GOLDEN


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_disc_masks_bit_exact():
    g = _load("masks.npz")
    centres = [tuple(c) for c in g["centres"]]
    for dim in g["dims"]:
        ref = np.unpackbits(g["mask_%d" % dim], axis=1)[:, : dim * dim].astype(bool)
        got = orc.disc_masks(centres, int(dim)).numpy()
        assert (got == ref).all(), dim
        # a disc of radius 0.2 covers ~12.5% of the square when fully inside
        assert abs(ref[0].mean() - np.pi * 0.04) < 0.03 or dim <= 12


def test_schedule_tables():
    g = _load("schedule.npz")
    for S in (50, 10):
        ts, a, ap, s1m = orc.ddim_alphas(S)
        assert (ts == g["t_%d" % S]).all()
        np.testing.assert_allclose(a, g["a_%d" % S], rtol=1e-6)
        np.testing.assert_allclose(ap, g["ap_%d" % S], rtol=1e-6)
        np.testing.assert_allclose(s1m, g["s1m_%d" % S], rtol=1e-6)
    assert orc.ddim_timesteps(50)[-1] == 981 and orc.ddim_timesteps(10)[-1] == 901


def _block_params(C, heads, seed):
    """The attn2 / norm2 weights of the golden block, rebuilt from the seed (same keys as the reference)."""
    from ldm.modules.attention import BasicTransformerBlock
    blk = BasicTransformerBlock(C, heads, C // heads, context_dim=768, checkpoint=False)
    cs = seeded_fill_(blk, seed)
    return blk, cs


@pytest.mark.parametrize("name", ["d40", "d80", "d160", "d8k4", "k0"])
def test_block_section_maps_and_dcoef(name):
    g = _load("block_%s.npz" % name)
    dim, C, heads, K, seed = (int(g[k]) for k in ("dim", "C", "heads", "K", "seed"))
    x, context, local_ctx = gi.block_inputs(dim, C, K, seed, gi.load_uncond())
    assert abs(gi.input_checksum(x, context, local_ctx) - float(g["input_checksum"])) < 1e-3 * float(g["input_checksum"]) * 1e-3 + 1e-2
    blk, cs = _block_params(C, heads, seed)
    assert abs(cs - float(g["checksum"])) <= 1e-6 * float(g["checksum"]), "seeded weight generator drifted"
    a2 = blk.attn2
    wq, wk, wv = a2.to_q.weight.detach(), a2.to_k.weight.detach(), a2.to_v.weight.detach()
    wo, bo = a2.to_out[0].weight.detach(), a2.to_out[0].bias.detach()
    with torch.no_grad():
        x1 = blk.attn1(blk.norm1(x)) + x                 # torch ops outside the hot path (self-attention)
        xn = blk.norm2(x1)
    masks = orc.disc_masks([tuple(c) for c in g["centres"]], dim)
    coef = torch.from_numpy(g["coef"])
    uncond = gi.load_uncond()

    # (1) reference form (K+1 batch-2 calls, blend after to_out) == what the reference left in x
    sec_ref = orc.block_xattn_reference_form(xn, context, local_ctx, uncond, masks, coef, wq, wk, wv, wo, bo, heads)
    np.testing.assert_allclose(sec_ref.numpy(), g["section"], rtol=0, atol=2e-4)

    # (2) fused form (pre-projection blend, single to_out) == the same thing
    q = xn @ wq.t()
    ctxs = torch.cat([context] + local_ctx)
    k, v = ctxs @ wk.t(), ctxs @ wv.t()
    scale = (C // heads) ** -0.5
    cg = coef.clone().requires_grad_(K > 0)
    fused, maps = orc.fused_xattn(q, k, v, masks, cg, heads, scale, want_maps=True)
    sec_fused = fused @ wo.t() + bo
    np.testing.assert_allclose(sec_fused.detach().numpy(), g["section"], rtol=0, atol=2e-4)

    # (3) attention maps of every (row, context) pair, at the stored pixels
    pix = torch.from_numpy(g["map_pixels"])
    np.testing.assert_allclose(maps[:, :, pix, :].detach().numpy(), g["maps"], rtol=0, atol=2e-6)

    # (4) d(0.5 sum out^2)/dcoef through the rest of the block, vs the reference's autograd
    if K:
        xo = sec_fused + x1
        out = blk.ff(blk.norm3(xo)) + xo
        for p in blk.parameters():
            p.requires_grad_(False)
        (0.5 * (out * out).sum()).backward()
        np.testing.assert_allclose(cg.grad.numpy(), g["dcoef"], rtol=2e-3)
        np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=0, atol=5e-4)
