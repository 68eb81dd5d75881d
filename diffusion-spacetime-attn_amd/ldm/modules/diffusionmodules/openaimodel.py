"""SD-v1 UNet (the `use_spatial_transformer=True, legacy=False` configuration of
configs/stable-diffusion/v1-inference.yaml) with the reference's hook surface:
`UNetModel.forward(x, text_index, timesteps, context, y, coef, bboxs_curr)` threads
`(context, timesteps[0], text_index, coef, bboxs_curr)` to every SpatialTransformer
(reference openaimodel.py:80-88, :710-743). Module and parameter names follow the reference's
state_dict (`input_blocks.N.M...`, `middle_block...`, `output_blocks...`, `time_embed`, `out`), so an
SD-v1-4 checkpoint loads without key remapping. Outside autograd on an NHWC trunk the 3x3 convolutions of the
ResBlocks and Upsample layers run on csrc/sta_conv.hip (implicit GEMM on the MFMAs; bias, skip add and the nearest-2x
read folded in); stride-2, 1x1 and conv_in / conv_out stay on MIOpen, as does everything under autograd.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ldm.modules.attention import SpatialTransformer
from ldm.modules.diffusionmodules.util import checkpoint, normalization, timestep_embedding, zero_module
from sta import fused as _fused


_conv3x3 = _fused.conv3x3_module


class TimestepBlock(nn.Module):
    """Marker: forward(x, emb)."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Feeds each child what it understands (reference openaimodel.py:74-88)."""

    def forward(self, x, emb, context=None, time=None, text_index=None, coef=None, bboxs_curr=None):
        if isinstance(x, tuple):      # (h, skip) of an output block, not concatenated: its first layer reads both in place
            x = self[0].forward_cat(x[0], x[1], emb)
            for layer in list(self)[1:]:
                if isinstance(layer, SpatialTransformer):
                    x = layer(x, context, time, text_index, coef=coef, bboxs_curr=bboxs_curr)
                elif isinstance(layer, TimestepBlock):
                    x = layer(x, emb)
                else:
                    x = layer(x)
            return x
        for layer in self:
            if isinstance(layer, SpatialTransformer):
                x = layer(x, context, time, text_index, coef=coef, bboxs_curr=bboxs_curr)
            elif isinstance(layer, TimestepBlock):
                x = layer(x, emb)
            else:
                x = layer(x)
        return x


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert dims == 2
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.conv = nn.Conv2d(self.channels, self.out_channels, 3, padding=padding)

    def forward(self, x):
        if self.use_conv and _fused.conv3x3_supported(x, self.conv.weight, up2=True):
            return _conv3x3(self, self.conv, x, bias=self.conv.bias, up2=True)      # the upsampled tensor is never written
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if self.use_conv and _fused.conv3x3_tracked_supported(x, self.conv.weight):
            return _fused.add_bias_tracked(_fused.conv3x3_tracked(self, self.conv, x), None, self.conv.bias)
        return self.conv(x) if self.use_conv else x


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert dims == 2
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if use_conv:
            self.op = nn.Conv2d(self.channels, self.out_channels, 3, stride=2, padding=padding)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        return self.op(x)


class ResBlock(TimestepBlock):
    """GroupNorm-SiLU-conv, + timestep embedding, GroupNorm-SiLU-conv, skip (reference :163-275;
    the scale-shift-norm and up/down variants are not used by SD-v1)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        if use_scale_shift_norm or up or down or dims != 2:
            raise NotImplementedError("SD-v1 ResBlock only")
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_checkpoint = use_checkpoint
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)

    def forward(self, x, emb):
        return checkpoint(self._forward, (x, emb), self.parameters(), self.use_checkpoint)

    def _forward(self, x, emb):
        if _fused.usable(x):
            return self._forward_fused(x, emb)
        if _fused.tracked_usable(x):
            return self._forward_tracked(x, emb)
        h = self.in_layers(x)
        h = h + self.emb_layers(emb).type(h.dtype)[:, :, None, None]
        return self.skip_connection(x) + self.out_layers(h)

    def cat_supported(self, xa, xb):
        sc = self.skip_connection
        return (isinstance(sc, nn.Conv2d) and sc.kernel_size == (1, 1) and not self.use_checkpoint
                and _fused.cat_supported(xa, xb, self.in_layers[0].num_groups, sc.weight))

    def forward_cat(self, xa, xb, emb):
        """_forward_fused of cat([xa, xb], dim=1) without the concatenated tensor (the output blocks' `th.cat([h, hs.pop()], dim=1)`,
        reference openaimodel.py:740): the first GroupNorm reads both tensors and writes the normalised concatenation, the 1x1 skip
        convolution is one row GEMM over both."""
        gn1, _, conv1 = self.in_layers
        gn2, _, _, conv2 = self.out_layers
        sc = self.skip_connection
        b, ca, hh, ww = xa.shape
        h = _fused.groupnorm_silu_cat(xa, xb, gn1.weight, gn1.bias, gn1.num_groups, gn1.eps)
        h = _conv3x3(self, conv1, h)
        add = self.emb_layers(emb).float() + conv1.bias.float()
        h = _fused.groupnorm_silu(h, gn2.weight, gn2.bias, gn2.num_groups, gn2.eps, add=add)
        w_sc = sc.weight[:, :, 0, 0]
        skip = _fused.linear_rows_cat(xa.permute(0, 2, 3, 1).reshape(b, hh * ww, ca), xb.permute(0, 2, 3, 1).reshape(b, hh * ww, xb.shape[1]),
                                      _fused.packed_linear_weight(self, sc, w_sc), w_sc.shape[0])
        skip = skip.view(b, hh, ww, -1).permute(0, 3, 1, 2)                            # NHWC view of [b, hw, c]
        return _conv3x3(self, conv2, h, bias=conv2.bias + sc.bias, res=skip)

    def _forward_tracked(self, x, emb):
        """Tracked epochs on an NHWC trunk (opt-in, sta.fused.TRACKED): the op structure of _forward_fused with every glue pass an
        autograd Function whose backward is one HIP input-gradient kernel (csrc/sta_unet_bwd.hip); convolutions through MIOpen."""
        gn1, _, conv1 = self.in_layers
        gn2, _, _, conv2 = self.out_layers
        conv = lambda cv, t: (_fused.conv3x3_tracked(self, cv, t) if _fused.conv3x3_tracked_supported(t, cv.weight)
                              else F.conv2d(t, cv.weight, None, cv.stride, cv.padding))      # forward + input gradient on csrc/sta_conv.hip
        h = _fused.groupnorm_silu_tracked(x, gn1.weight, gn1.bias, gn1.num_groups, gn1.eps)
        h = conv(conv1, h)
        add = (self.emb_layers(emb).float() + conv1.bias.float()).detach()            # no path from the blend weights to the embedding
        h = _fused.groupnorm_silu_tracked(h, gn2.weight, gn2.bias, gn2.num_groups, gn2.eps, add=add)
        h = conv(conv2, h)
        skip, bias = x, conv2.bias
        if not isinstance(self.skip_connection, nn.Identity):
            sc = self.skip_connection
            skip = F.conv2d(x, sc.weight, None, sc.stride, sc.padding)
            bias = conv2.bias + sc.bias
        return _fused.add_bias_tracked(skip, h, bias)

    def _forward_fused(self, x, emb):
        """Inference: GroupNorm+SiLU are one pass each; the timestep-embedding add AND conv1's bias ride into the
        second GroupNorm pass as a per-(b, c) pre-add; conv2's bias, the skip's bias and the residual add are the
        epilogue of the second convolution (csrc/sta_conv.hip; with the library convolution: one pass of csrc/sta_unet.hip)."""
        gn1, _, conv1 = self.in_layers
        gn2, _, _, conv2 = self.out_layers
        h = _fused.groupnorm_silu(x, gn1.weight, gn1.bias, gn1.num_groups, gn1.eps)
        h = _conv3x3(self, conv1, h)
        add = self.emb_layers(emb).float() + conv1.bias.float()                       # [B, C_out]
        h = _fused.groupnorm_silu(h, gn2.weight, gn2.bias, gn2.num_groups, gn2.eps, add=add)
        skip, bias = x, conv2.bias
        if not isinstance(self.skip_connection, nn.Identity):
            sc = self.skip_connection
            skip = F.conv2d(x, sc.weight, None, sc.stride, sc.padding)
            bias = conv2.bias + sc.bias
        return _conv3x3(self, conv2, h, bias=bias, res=skip)


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True):
        super().__init__()
        if not use_spatial_transformer or context_dim is None:
            raise NotImplementedError("only the cross-attention (SpatialTransformer) UNet of SD-v1 is provided")
        if num_classes is not None or n_embed is not None or resblock_updown or use_scale_shift_norm:
            raise NotImplementedError("option not used by SD-v1")
        if num_heads == -1 and num_head_channels == -1:
            raise ValueError("Either num_heads or num_head_channels has to be set")
        context_dim = list(context_dim)[0] if isinstance(context_dim, (list, tuple)) else context_dim
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.channel_mult = num_res_blocks, list(attention_resolutions), list(channel_mult)
        self.use_checkpoint, self.num_heads, self.num_head_channels = use_checkpoint, num_heads, num_head_channels
        self.num_classes = None
        self.dtype = torch.float16 if use_fp16 else torch.float32

        def heads_for(ch):
            return (num_heads, ch // num_heads) if num_head_channels == -1 else (ch // num_head_channels, num_head_channels)

        def res(cin, cout):
            return ResBlock(cin, emb_dim, dropout, out_channels=cout, use_checkpoint=use_checkpoint)

        def attn(ch):
            h, dh = heads_for(ch)
            return SpatialTransformer(ch, h, dh, depth=transformer_depth, context_dim=context_dim)

        emb_dim = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, emb_dim), nn.SiLU(), nn.Linear(emb_dim, emb_dim))

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        skip_chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(self.channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in self.attention_resolutions:
                    layers.append(attn(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                skip_chans.append(ch)
            if level != len(self.channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, out_channels=ch)))
                skip_chans.append(ch)
                ds *= 2

        self.middle_block = TimestepEmbedSequential(res(ch, ch), attn(ch), res(ch, ch))

        self.output_blocks = nn.ModuleList()
        for level, mult in reversed(list(enumerate(self.channel_mult))):
            for i in range(num_res_blocks + 1):
                layers = [res(ch + skip_chans.pop(), model_channels * mult)]
                ch = model_channels * mult
                if ds in self.attention_resolutions:
                    layers.append(attn(ch))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))

        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(nn.Conv2d(model_channels, out_channels, 3, padding=1)))

    def forward(self, x, text_index=None, timesteps=None, context=None, y=None, coef=None, bboxs_curr=None, **kwargs):
        if y is not None:
            raise ValueError("this UNet is not class-conditional")
        wdtype = self.time_embed[0].weight.dtype
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels).to(wdtype))
        time = timesteps[0]
        h, skips = x.to(wdtype), []
        if self.input_blocks[0][0].weight.is_contiguous(memory_format=torch.channels_last) and h.is_cuda:
            h = h.contiguous(memory_format=torch.channels_last)       # NHWC pipeline (sta.pipeline.build_sd_v1)
        for module in self.input_blocks:
            h = module(h, emb, context, time, text_index, coef=coef, bboxs_curr=bboxs_curr)
            skips.append(h)
        h = self.middle_block(h, emb, context, time, text_index, coef=coef, bboxs_curr=bboxs_curr)
        for module in self.output_blocks:
            sk = skips.pop()
            if isinstance(module[0], ResBlock) and module[0].cat_supported(h, sk):
                h = module((h, sk), emb, context, time, text_index, coef=coef, bboxs_curr=bboxs_curr)      # the concatenation is read in place
            else:
                h = module(torch.cat([h, sk], dim=1), emb, context, time, text_index, coef=coef, bboxs_curr=bboxs_curr)
        if _fused.usable(h):
            gn, _, conv = self.out
            h = _fused.groupnorm_silu(h, gn.weight, gn.bias, gn.num_groups, gn.eps)
            return conv(h).to(x.dtype).contiguous()
        if _fused.tracked_usable(h):
            gn, _, conv = self.out
            h = _fused.groupnorm_silu_tracked(h, gn.weight, gn.bias, gn.num_groups, gn.eps)
            return conv(h).to(x.dtype)
        return self.out(h).to(x.dtype)

    def transformer_blocks(self):
        from ldm.modules.attention import BasicTransformerBlock
        return [m for m in self.modules() if isinstance(m, BasicTransformerBlock)]
