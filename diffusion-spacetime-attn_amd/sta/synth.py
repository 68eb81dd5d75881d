"""Deterministic synthetic weights (there is no network: no SD-v1-4 / CLIP checkpoints on disk).

`seeded_fill_` gives every tensor of a state_dict a value that depends only on (seed, key name,
shape), so two implementations with the same state_dict contract (the reference's modules and
ours) can be given identical weights without shipping them: fixtures then hold inputs and outputs
only. Zero-initialised layers of the reference (`SpatialTransformer.proj_out`, the UNet's `out`
conv, ResBlock out convs) are re-randomised on purpose — with them at zero the cross-attention
path contributes exactly nothing to the output.
"""
import zlib

import numpy as np
import torch


def _tensor_for(name, shape, seed):
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    n = rng.standard_normal(size=tuple(shape), dtype=np.float32)
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) >= 2:                       # Linear / Conv weight: variance-preserving
        fan_in = int(np.prod(shape[1:]))
        return n * (1.0 / np.sqrt(fan_in))
    if leaf == "weight":                      # norm gains
        return 1.0 + 0.1 * n
    return 0.05 * n                           # biases


def seeded_tensor(tag, shape, seed, scale=1.0):
    """float32 standard-normal tensor that depends only on (seed, tag, shape) — for test inputs that
    both the golden generator and the tests rebuild instead of storing."""
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(tag.encode())]))
    return torch.from_numpy(rng.standard_normal(size=tuple(shape), dtype=np.float32) * np.float32(scale))


def seeded_fill_(module, seed=0):
    """In-place: fill every parameter/buffer of `module` from (seed, name). Returns a checksum."""
    total = 0.0
    with torch.no_grad():
        for name, t in sorted(module.state_dict().items()):
            if not t.dtype.is_floating_point:
                continue
            val = torch.from_numpy(_tensor_for(name, t.shape, seed))
            t.copy_(val.to(t.dtype))
            total += float(val.double().abs().sum())
    return total


def device_fill_(module, seed=0):
    """Fast on-device variant for benchmarking (values differ from seeded_fill_)."""
    g = torch.Generator(device=next(module.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for name, t in sorted(module.state_dict().items()):
            if not t.dtype.is_floating_point:
                continue
            n = torch.randn(t.shape, generator=g, device=t.device, dtype=torch.float32)
            leaf = name.rsplit(".", 1)[-1]
            if t.dim() >= 2:
                n = n / float(np.sqrt(np.prod(t.shape[1:])))
            elif leaf == "weight":
                n = 1.0 + 0.1 * n
            else:
                n = 0.05 * n
            t.copy_(n.to(t.dtype))


def calibrate_decoder_(model, latent, target_std=0.25):
    """Synthetic VAE weights decode a sampled latent to values far outside [-1, 1]: the image clamp of plms.py:249-250 then
    saturates everywhere and the fidelity loss has an exactly-zero gradient. Rescale the decoder's last convolution so that
    the decoded `latent` (one already-sampled x0) has mean 0 and the given std — (img + 1) / 2 stays inside (0, 1) almost
    everywhere, as it does with real weights. Benchmarks / timing runs with synthetic weights only. Returns (mean, std) before."""
    conv = model.first_stage_model.decoder.conv_out
    with torch.no_grad():
        img = model.decode_first_stage(latent.to(conv.weight.dtype)).float()
        mean, std = float(img.mean()), float(img.std())
        s = target_std / max(std, 1e-12)
        conv.weight.mul_(s)
        conv.bias.sub_(mean).mul_(s)       # decode is affine in (weight, bias) of its last layer: out -> (out - mean) * s
    return mean, std


class SyntheticCLIP(torch.nn.Module):
    """Stand-in for the third-party CLIP ViT-B/32 the reference's fidelity loss calls (plms.py:21-45):
    same interface (`encode_image([1,3,224,224])`, `encode_text(str)`, 512-d features), tiny frozen
    random weights. It exists so that the gradient path loss -> VAE decoder -> 51 UNet calls -> blend
    weights can be exercised and timed without CLIP weights; it says nothing about image quality."""

    def __init__(self, dim=512, seed=0):
        super().__init__()
        self.patch = torch.nn.Conv2d(3, 256, kernel_size=32, stride=32)
        self.proj = torch.nn.Linear(256, dim)
        self.dim = dim
        seeded_fill_(self, seed)
        for p in self.parameters():
            p.requires_grad_(False)

    def encode_image(self, image):
        w = self.patch.weight
        f = self.patch(image.to(w.dtype)).flatten(2).mean(-1)
        return self.proj(torch.nn.functional.gelu(f))

    def encode_text(self, text):
        rng = np.random.Generator(np.random.PCG64([7, zlib.crc32(str(text).encode("utf-8"))]))
        return torch.from_numpy(rng.standard_normal((1, self.dim), dtype=np.float32)).to(self.proj.weight.device)


def synthetic_clip(device="cuda"):
    """`--clip sta.synth:synthetic_clip`: the stand-in above as a loss-model factory (ldm...plms.load_clip_model)."""
    return SyntheticCLIP().to(device)
