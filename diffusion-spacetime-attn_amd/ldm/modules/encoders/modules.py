"""Text encoders producing the [B, 77, 768] contexts the cross-attention consumes.

`FrozenCLIPEmbedder` mirrors the reference's wrapper around HF `openai/clip-vit-large-patch14`
(reference ldm/modules/encoders/modules.py:137-162); it needs the weights locally (no network here).
`SyntheticTextEmbedder` is the stand-in used when they are absent (benchmarks, tests): a
deterministic embedding per string with the reference fixture's scale (mean |x| of
uncond_fix_radius_0p2_g0.pt is 0.777). It is data, not a model: the third-party CLIP arithmetic is
out of scope and parity-unpinned.
"""
import zlib

import numpy as np
import torch
from torch import nn


class SyntheticTextEmbedder(nn.Module):
    def __init__(self, max_length=77, dim=768, scale=0.975):
        super().__init__()
        self.max_length, self.dim, self.scale = max_length, dim, scale
        self.register_buffer("_anchor", torch.zeros(1), persistent=False)

    @property
    def device(self):
        return self._anchor.device

    def encode(self, text):
        texts = [text] if isinstance(text, str) else list(text)
        out = []
        for t in texts:
            rng = np.random.Generator(np.random.PCG64([20230817, zlib.crc32(t.encode("utf-8"))]))
            out.append(rng.standard_normal((self.max_length, self.dim), dtype=np.float32) * np.float32(self.scale))
        return torch.from_numpy(np.stack(out)).to(self.device)

    forward = encode


class FrozenCLIPEmbedder(nn.Module):
    """HF CLIP text transformer, frozen; output = last_hidden_state for 77 padded tokens."""

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77):
        super().__init__()
        from transformers import CLIPTextModel, CLIPTokenizer
        try:
            self.tokenizer = CLIPTokenizer.from_pretrained(version, local_files_only=True)
            self.transformer = CLIPTextModel.from_pretrained(version, local_files_only=True)
        except Exception as e:   # no weights on disk and no network
            raise RuntimeError("CLIP text encoder weights for %s are not available locally; use "
                               "SyntheticTextEmbedder or provide the HF cache" % version) from e
        self.device_name, self.max_length = device, max_length
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, text):
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        tokens = enc["input_ids"].to(next(self.transformer.parameters()).device)
        return self.transformer(input_ids=tokens).last_hidden_state

    def encode(self, text):
        return self(text)
