"""ORACLE tooling — seeded inputs shared by oracle/gen_golden.py (writer) and tests/ (readers).

Random inputs are rebuilt from (seed, tag) on both sides instead of being stored; the fixtures keep
their checksum so a generator drift shows up as a checksum error, not as a parity failure.
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffusion-spacetime-attn_amd"))
from sta.synth import seeded_tensor  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")
LAT = 32   # latent side of the UNet-level fixtures


def load_uncond():
    """CLIP-L/14 embedding of "" — the reference's only tensor fixture (uncond_fix_radius_0p2_g0.pt)."""
    return torch.from_numpy(np.load(os.path.join(GOLDEN, "uncond_clip_l14.npy")))


def block_inputs(dim, C, K, seed, uncond):
    x = seeded_tensor("x", (2, dim * dim, C), seed)
    context = torch.cat([uncond + seeded_tensor("uc_noise", (1, 77, 768), seed, 0.01), seeded_tensor("c", (1, 77, 768), seed, 0.78)])
    local_ctx = [seeded_tensor("c%d" % i, (1, 77, 768), seed, 0.78) for i in range(K)]
    return x, context, local_ctx


def input_checksum(x, context, local_ctx):
    return float(x.double().abs().sum() + context.double().abs().sum() + sum(c.double().abs().sum() for c in local_ctx))


def unet_inputs(K, seed, lat=LAT):
    c = seeded_tensor("c", (1, 77, 768), seed, 0.78)
    local_ctx = [seeded_tensor("c%d" % i, (1, 77, 768), seed, 0.78) for i in range(K)]
    x = seeded_tensor("x", (1, 4, lat, lat), seed)
    return c, local_ctx, x


def loss_image(seed, side=512):
    """A [3, side, side] image in [0, 1] for the loss front-end fixture: smooth blobs + noise (so that resizing and
    pooling are not no-ops), rebuilt from the seed on both sides."""
    low = seeded_tensor("img_low", (1, 3, side // 32, side // 32), seed)
    up = torch.nn.functional.interpolate(low, size=(side, side), mode="bicubic", align_corners=False)[0]
    return torch.sigmoid(up + 0.5 * seeded_tensor("img_noise", (3, side, side), seed))
