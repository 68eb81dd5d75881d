"""fp8 (OCP e4m3fn) weights for the Linear layers of the transformer blocks — BASELINE configs[4]
("fp8 UNet weights on CDNA4 fp8 MFMA").

Which layers (reference ldm/modules/attention.py): attn1.to_q/to_k/to_v and both to_out (:164-171), GEGLU.proj
(:50) and the FeedForward output Linear (:69) — 6 of the 7 GEMMs of a block. attn2.to_q/to_k/to_v stay 16 bit: they
produce the q/K/V of the spatial-temporal cross-attention, whose per-step maps are held to 1e-3 (8 mantissa bits
are already too few there), and to_k/to_v run once per prompt.

How: weight rows are quantised once (`Fp8Linear.from_linear`: one fp32 scale per OUTPUT channel, absmax / 448),
activations per call by the HIP kernel `sta_quant_rows_fp8` (one fp32 scale per row, csrc/sta_fp8.hip); the GEMM is
e4m3 x e4m3 -> fp32 -> 16 bit by hipBLASLt's row-scaled e4m3 GEMM (`torch._scaled_mm`: a plain library GEMM, no own fp8 kernel) with both
scale vectors applied to the fp32 accumulators. Weight bytes halve (UNet transformer Linears: 0.61 GB -> 0.31 GB);
on gfx950 the non-block-scaled fp8 MFMA runs at the 16-bit rate, so this is a weight-MEMORY option, not a rate option.
Inference only (no autograd through the quantiser)."""
import torch
from torch import nn

from . import lib as _lib

F8 = torch.float8_e4m3fn
E4M3_MAX = 448.0
_DT = {torch.bfloat16: _lib.STA_BF16, torch.float16: _lib.STA_F16}


def quant_rows(x2d):
    """x [R, C] 16-bit CUDA -> (xq [R, C] e4m3, scale [R, 1] fp32) through csrc/sta_fp8.hip."""
    if not x2d.is_cuda or x2d.dtype not in _DT:
        raise RuntimeError("fp8 activation quantisation needs 16-bit CUDA/HIP tensors (there is no CPU path)")
    x2d = x2d.contiguous()
    R, C = x2d.shape
    xq = torch.empty((R, C), dtype=F8, device=x2d.device)
    scale = torch.empty((R, 1), dtype=torch.float32, device=x2d.device)
    _lib.check(_lib.load().sta_quant_rows_fp8(x2d.data_ptr(), xq.data_ptr(), scale.data_ptr(), R, C, _DT[x2d.dtype],
                                              torch.cuda.current_stream(x2d.device).cuda_stream), "sta_quant_rows_fp8")
    return xq, scale


def scaled_mm(a_q, b_q, scale_a, scale_b, bias, out_dtype):
    """e4m3 x e4m3 -> fp32 accumulators x (row scale x column scale) -> 16 bit. hipBLASLt's row-wise-scaled fp8 GEMM
    writes bf16 only (ROCm 7.2 / torch 2.10: "rowwise _scaled_mm only supports BFloat16 output"), so an fp16 model takes
    the bf16 result through one cast pass; values are GEMM outputs of O(1..100), far inside both ranges."""
    out = torch._scaled_mm(a_q, b_q, scale_a=scale_a, scale_b=scale_b, bias=None if bias is None else bias.to(torch.bfloat16),
                           out_dtype=torch.bfloat16)
    return out if out_dtype == torch.bfloat16 else out.to(out_dtype)


def quant_weight(w):
    """w [out, in] -> (wq [out, in] e4m3, scale [1, out] fp32): one scale per output channel."""
    amax = w.detach().float().abs().amax(dim=1, keepdim=True).clamp_min(1e-12)
    scale = amax / E4M3_MAX
    return (w.detach().float() / scale).to(F8), scale.t().contiguous()


class Fp8Linear(nn.Module):
    """Drop-in for nn.Linear (same call signature and `in_features`/`out_features`), e4m3 weight + fp32 channel scales."""

    def __init__(self, wq, wscale, bias):
        super().__init__()
        self.out_features, self.in_features = wq.shape
        self.register_buffer("weight_q", wq)
        self.register_buffer("weight_scale", wscale)
        self.bias = None if bias is None else nn.Parameter(bias.detach().clone(), requires_grad=False)

    @classmethod
    def from_linear(cls, lin):
        return cls(*quant_weight(lin.weight), lin.bias)

    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            raise RuntimeError("Fp8Linear is an inference layer (BASELINE configs[4] has fixed blend weights)")
        xq, sx = quant_rows(x.reshape(-1, self.in_features))
        out = scaled_mm(xq, self.weight_q.t(), sx, self.weight_scale, self.bias, x.dtype)
        return out.view(*x.shape[:-1], self.out_features)

    def forward_transposed(self, xq, sx, out_dtype):
        """W . x^T for an already quantised x: [out, R] (the self-attention kernel wants V transposed)."""
        return scaled_mm(self.weight_q, xq.t(), self.weight_scale.t().contiguous(), sx.t().contiguous(), None, out_dtype)


def convert_transformer_linears_(unet):
    """In place: the Linear layers listed in the module docstring of every BasicTransformerBlock of `unet` become
    Fp8Linear. Returns (number converted, 16-bit bytes before, bytes after)."""
    from ldm.modules.attention import BasicTransformerBlock
    n = before = after = 0
    for blk in unet.modules():
        if not isinstance(blk, BasicTransformerBlock):
            continue
        sites = [(blk.attn1, "to_q"), (blk.attn1, "to_k"), (blk.attn1, "to_v"), (blk.attn1.to_out, "0"), (blk.attn2.to_out, "0"),
                 (blk.ff.net[0], "proj"), (blk.ff.net, "2")]
        for parent, name in sites:
            lin = getattr(parent, name)
            if isinstance(lin, nn.Linear):
                q = Fp8Linear.from_linear(lin)
                before += lin.weight.numel() * lin.weight.element_size()
                after += q.weight_q.numel() + q.weight_scale.numel() * 4
                setattr(parent, name, q)
                n += 1
    return n, before, after
