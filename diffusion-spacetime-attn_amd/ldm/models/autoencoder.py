"""KL-VAE decoder of SD-v1 (first_stage_model) — the image end of the loss path
(reference ldm/models/autoencoder.py:330-333, ldm/modules/diffusionmodules/model.py:462-560).
Stays on PyTorch-ROCm/MIOpen; parameter names follow the SD-v1-4 state_dict
(`first_stage_model.post_quant_conv`, `first_stage_model.decoder.*`). Encoder is not needed for
text-to-image sampling and is not provided."""
import torch
import torch.nn.functional as F
from torch import nn

from ldm.modules.diffusionmodules.util import Normalize
from sta import fused as _fused


def _norm_silu(norm, x):
    """GroupNorm + SiLU: one pass through csrc/sta_unet.hip outside autograd (the image decode of fixed-weight
    sampling and of the last epoch), the eager pair otherwise (the loss path of the tracked epochs)."""
    if _fused.usable(x):
        return _fused.groupnorm_silu(x, norm.weight, norm.bias, norm.num_groups, norm.eps)
    if _fused.tracked_usable(x):      # NHWC decoder under autograd (tracked epochs, sta.fused.tracked): HIP input gradient
        return _fused.groupnorm_silu_tracked(x, norm.weight, norm.bias, norm.num_groups, norm.eps)
    return F.silu(norm(x))


class ResnetBlock(nn.Module):
    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1)

    def forward(self, x, gn_next=True):
        """`gn_next=False`: the result feeds an Upsample convolution, not a GroupNorm — its producer keeps no statistics."""
        if _fused.conv3x3_supported(x, self.conv1.weight):
            # fixed-weight decode on an NHWC decoder: both convolutions on csrc/sta_conv.hip, the shortcut add in the second one's epilogue
            h = _fused.conv3x3_module(self, self.conv1, _norm_silu(self.norm1, x), bias=self.conv1.bias)
            skip = self.nin_shortcut(x) if hasattr(self, "nin_shortcut") else x
            return _fused.conv3x3_module(self, self.conv2, _norm_silu(self.norm2, h), bias=self.conv2.bias, res=skip, stats=gn_next)
        h = self.conv1(_norm_silu(self.norm1, x))
        h = self.conv2(_norm_silu(self.norm2, h))
        return (self.nin_shortcut(x) if hasattr(self, "nin_shortcut") else x) + h


class AttnBlock(nn.Module):
    """Single-head spatial self-attention with 1x1-conv projections."""

    def __init__(self, channels):
        super().__init__()
        self.norm = Normalize(channels)
        self.q = nn.Conv2d(channels, channels, 1)
        self.k = nn.Conv2d(channels, channels, 1)
        self.v = nn.Conv2d(channels, channels, 1)
        self.proj_out = nn.Conv2d(channels, channels, 1)

    def forward(self, x):
        b, c, h, w = x.shape
        if _fused.tracked_usable(x):
            n = _fused.groupnorm_silu_tracked(x, self.norm.weight, self.norm.bias, self.norm.num_groups, self.norm.eps, silu=False)
        else:
            n = self.norm(x)
        q, k, v = (m(n).flatten(2).transpose(1, 2) for m in (self.q, self.k, self.v))   # [b, hw, c]
        if not x.is_cuda:
            o = F.scaled_dot_product_attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1), scale=c ** -0.5).squeeze(1)
        else:
            o = _attention_chunked(q, k, v, c ** -0.5)
        return x + self.proj_out(o.transpose(1, 2).reshape(b, c, h, w))


ATTN_CHUNK = 8      # images per batched GEMM of the decoder's attention


def _attention_chunked(q, k, v, scale):
    """softmax(scale q k^T) v for the decoder's one single-head attention (hw = 4096, c = 512; reference
    ldm/modules/diffusionmodules/model.py AttnBlock: bmm - softmax - bmm) as plain library GEMMs with an fp32 softmax, a few
    images at a time: no Triton-built SDPA kernel in the product path. The scale rides on q so that the 16-bit
    scores stay small; runs once per image, under autograd in the tracked epochs (P of a chunk is what backward keeps)."""
    # CONTIGUOUS [b, hw, c] operands: q, k, v arrive as transposed views of the 1x1 convolutions' [b, c, hw] outputs, and a
    # batched GEMM whose operands are such views returns wrong values in bf16 on this library build (max error 5.7 against
    # fp32 at [32..48, 4096, 512]; the same call on contiguous tensors is right: tools/repro_bmm_fault.py, profiles/r04_bmm_repro.txt)
    # — the memory-access fault the 32-image bf16 decode ran into. Three 4 MB-per-image copies, once per image.
    q, k, v = (q * scale).contiguous(), k.contiguous(), v.contiguous()
    outs = []
    for i in range(0, q.shape[0], ATTN_CHUNK):
        s = torch.bmm(q[i:i + ATTN_CHUNK], k[i:i + ATTN_CHUNK].transpose(1, 2))
        p = torch.softmax(s.float(), dim=-1).to(q.dtype)
        outs.append(torch.bmm(p, v[i:i + ATTN_CHUNK]))
    return outs[0] if len(outs) == 1 else torch.cat(outs)


class _Up(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        if _fused.conv3x3_supported(x, self.conv.weight, up2=True):
            return _fused.conv3x3_module(self, self.conv, x, bias=self.conv.bias, up2=True)      # the upsampled tensor is never written
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Decoder(nn.Module):
    def __init__(self, *, ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=(), dropout=0.0,
                 in_channels=3, resolution=256, z_channels=4, double_z=True, **ignored):
        super().__init__()
        if attn_resolutions:
            raise NotImplementedError("SD-v1 decoder has no per-level attention")
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in)
        self.up = nn.ModuleList()
        levels = []
        for i_level in reversed(range(self.num_resolutions)):
            lvl = nn.Module()
            lvl.block = nn.ModuleList()
            lvl.attn = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                lvl.block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            if i_level != 0:
                lvl.upsample = _Up(block_in)
            levels.insert(0, lvl)
        self.up.extend(levels)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            for j, blk in enumerate(lvl.block):
                h = blk(h, gn_next=not (i_level != 0 and j == len(lvl.block) - 1))      # the level's last block feeds the Upsample convolution
            if i_level != 0:
                h = lvl.upsample(h)
        return self.conv_out(_norm_silu(self.norm_out, h))


class AutoencoderKL(nn.Module):
    """Decode-only AutoencoderKL: `decode(z) = decoder(post_quant_conv(z))` (reference autoencoder.py:330-333)."""

    def __init__(self, ddconfig=None, embed_dim=4, lossconfig=None, **ignored):
        super().__init__()
        ddconfig = dict(ddconfig or dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                                         ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0))
        self.decoder = Decoder(**ddconfig)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)

    def decode(self, z):
        w = self.decoder.conv_in.weight
        if z.is_cuda and not w.is_contiguous() and w.is_contiguous(memory_format=torch.channels_last):
            z = z.contiguous(memory_format=torch.channels_last)       # NHWC decoder (sta.pipeline.set_recompute, per-call policy)
        return self.decoder(self.post_quant_conv(z))
