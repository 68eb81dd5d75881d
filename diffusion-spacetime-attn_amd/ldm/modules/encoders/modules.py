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
    """HF CLIP text transformer, frozen; output = last_hidden_state for 77 padded tokens
    (reference ldm/modules/encoders/modules.py:137-162).

    `from_config=True` builds the ViT-L/14 text tower from its architecture constants WITHOUT weights (they
    then come from the `cond_stage_model.transformer.*` keys of the SD-v1-4 checkpoint, which is how
    sta.pipeline.build_sd_v1 uses it); otherwise the weights are read from the local HF cache. The tokenizer's
    vocabulary files must be on disk either way (`tokenizer_path` or the HF cache): there is no network here, and a
    missing tokenizer raises instead of conditioning the UNet on something else."""

    VIT_L14_TEXT = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                        num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu",
                        layer_norm_eps=1e-5, projection_dim=768)

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, from_config=False,
                 tokenizer_path=None):
        super().__init__()
        from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
        try:
            self.tokenizer = CLIPTokenizer.from_pretrained(tokenizer_path or version, local_files_only=True)
        except Exception as e:   # no vocabulary on disk and no network
            raise RuntimeError("CLIP tokenizer files for %s are not available locally (pass tokenizer_path= / "
                               "--clip_tokenizer, or provide the HF cache); refusing to fall back to synthetic text "
                               "embeddings for a real checkpoint" % (tokenizer_path or version)) from e
        if from_config:
            self.transformer = CLIPTextModel(CLIPTextConfig(**self.VIT_L14_TEXT))
        else:
            try:
                self.transformer = CLIPTextModel.from_pretrained(version, local_files_only=True)
            except Exception as e:
                raise RuntimeError("CLIP text encoder weights for %s are not available locally" % version) from e
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
