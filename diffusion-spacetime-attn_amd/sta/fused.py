"""Python face of include/sta_unet.h: one-pass HIP kernels for the elementwise / normalisation chains of the
UNet trunk around the cross-attention path (ResBlock, SpatialTransformer, BasicTransformerBlock, GEGLU).

The forward kernels are used for inference: `usable()` is False whenever autograd is recording, and the modules then run
the reference's eager sequence (which is also what the CPU tests exercise). `fused.ENABLED = False` (tools/, tests) switches
them off. For the tracked (weight-optimisation) epochs the same forward kernels are wrapped in autograd Functions whose
backward is a HIP input-gradient kernel (csrc/sta_unet_bwd.hip: parameters are frozen, only d(input) exists): `tracked_usable()`,
`GroupNormSiLUFn`, `GegluFn`, `AddLayerNormFn`, `AddBiasFn`. They are opt-in (`with fused.tracked():`, entered by PLMSSampler
around a tracked epoch under the per-call recomputation policy of sta.pipeline.set_recompute) so that the other policies keep
the eager chain the goldens pin."""

import torch

from . import lib

_DT = {torch.bfloat16: lib.STA_BF16, torch.float16: lib.STA_F16}
ENABLED = True          # A/B switch for tools/ and tests; nothing reads the environment
_hold = 0               # > 0 inside `held()`: the forward of a block that will be re-run under autograd
TRACKED = False         # differentiable fused glue ops while autograd records (see module docstring)


class held:
    """Context in which the inference-only kernels (this module, HIP self-attention) are not used although
    autograd is off: the no-grad forward of a recomputed block must run the SAME op chain as its re-run in
    backward, otherwise gradients are taken at activations the forward never produced (util._Recompute)."""

    def __enter__(self):
        global _hold
        _hold += 1

    def __exit__(self, *exc):
        global _hold
        _hold -= 1


class tracked:
    """Context in which the differentiable fused glue ops are used while autograd records (tracked epochs)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global TRACKED
        self.prev, TRACKED = TRACKED, self.on
        return self

    def __exit__(self, *exc):
        global TRACKED
        TRACKED = self.prev


def is_nhwc(x):
    """A 4-D activation stored channels_last (NHWC in memory) and not also plain-contiguous."""
    return x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)


def usable(x):
    """Fused trunk kernels apply to dense (NCHW- or NHWC-contiguous) 16-bit CUDA activations outside autograd."""
    return (x.is_cuda and x.dtype in _DT and not torch.is_grad_enabled() and (x.is_contiguous() or is_nhwc(x))
            and ENABLED and not _hold)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def groupnorm_silu(x, weight, bias, groups, eps, add=None, silu=True):
    """act(GroupNorm(x + add[:, :, None, None])); x [B, C, *spatial] contiguous, add [B, C] or None."""
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    if add is not None:
        add = add.float().contiguous()
        assert add.shape == (B, C)
    def eager():                                # shapes outside the kernels' limits: the reference's op chain
        h = x if add is None else x + add.to(x.dtype)[:, :, None, None]
        h = torch.nn.functional.group_norm(h, groups, weight, bias, eps)
        return torch.nn.functional.silu(h) if silu else h
    y = torch.empty_like(x)                     # keeps the memory format
    if is_nhwc(x):
        L = lib.load()
        if C % 8 or (C // groups < 8 and C // groups != 4) or C > 4096 or groups > 64:
            return eager()
        st = _producer_stats(x) if GN_STATS_FROM_PRODUCER else None
        if st is not None:
            # the kernel that wrote x accumulated every channel's sum and sum of squares (sta_conv3x3_nhwc / sta_linear_rows_stats): one pass
            lib.check(L.sta_groupnorm_silu_nhwc_cstats(x.data_ptr(), None, C, st.data_ptr(), None, _ptr(add), weight.data_ptr(), bias.data_ptr(),
                                                       y.data_ptr(), B, C, HW, groups, float(eps), int(bool(silu)), _DT[x.dtype], _stream()),
                      "sta_groupnorm_silu_nhwc_cstats")
            return y
        ws = torch.empty(L.sta_groupnorm_nhwc_workspace_bytes(B, HW, groups) // 4, dtype=torch.float32, device=x.device)
        lib.check(L.sta_groupnorm_silu_nhwc(x.data_ptr(), _ptr(add), weight.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                            ws.data_ptr(), B, C, HW, groups, float(eps), int(bool(silu)), _DT[x.dtype], _stream()),
                  "sta_groupnorm_silu_nhwc")
        return y
    if HW % 8 or C % groups:
        return eager()
    lib.check(lib.load().sta_groupnorm_silu(x.data_ptr(), _ptr(add), weight.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                            B, C, HW, groups, float(eps), int(bool(silu)), _DT[x.dtype], _stream()),
              "sta_groupnorm_silu")
    return y


def geglu(h):
    """h [..., 2D] -> h[..., :D] * gelu(h[..., D:])."""
    D = h.shape[-1] // 2
    if D % 8:
        a, gate = h.chunk(2, dim=-1)
        return a * torch.nn.functional.gelu(gate)
    y = torch.empty(*h.shape[:-1], D, dtype=h.dtype, device=h.device)
    lib.check(lib.load().sta_geglu(h.data_ptr(), y.data_ptr(), h.numel() // (2 * D), D, _DT[h.dtype], _stream()), "sta_geglu")
    return y


def add_layernorm(x, f, bias, ln_weight, ln_bias, eps, store_sum=True, qfrag=False):
    """s = x + f + bias; returns (s, LayerNorm(s)); f / bias may be None; s is None if store_sum is False.
    `qfrag`: y comes back in QUERY-FRAGMENT order (sta_add_layernorm_qfrag: same shape and bytes, 16-row groups as 1-KiB MFMA
    operand fragments) for sta.ops.xattn_forward_proj(..., qfrag=True), its only legal consumer."""
    C = x.shape[-1]
    if qfrag:
        if C % 32 or C > 1024 or (x.numel() // C) % 16:
            raise ValueError("query-fragment order needs C %% 32 == 0, C <= 1024 and a multiple of 16 rows, got %s" % (tuple(x.shape),))
        s = torch.empty_like(x) if store_sum else None
        y = torch.empty_like(x)
        lib.check(lib.load().sta_add_layernorm_qfrag(x.data_ptr(), _ptr(f), _ptr(bias), ln_weight.data_ptr(), ln_bias.data_ptr(),
                                                     _ptr(s), y.data_ptr(), x.numel() // C, C, float(eps), _DT[x.dtype], _stream()),
                  "sta_add_layernorm_qfrag")
        return s, y
    if C % 8 or C > 2048:
        t = x if f is None else x + f
        t = t if bias is None else t + bias
        return (t if store_sum else None), torch.nn.functional.layer_norm(t, (C,), ln_weight, ln_bias, eps)
    s = torch.empty_like(x) if store_sum else None
    y = torch.empty_like(x)
    lib.check(lib.load().sta_add_layernorm(x.data_ptr(), _ptr(f), _ptr(bias), ln_weight.data_ptr(), ln_bias.data_ptr(),
                                           _ptr(s), y.data_ptr(), x.numel() // C, C, float(eps), _DT[x.dtype], _stream()),
              "sta_add_layernorm")
    return s, y


FRAG_XATTN, FRAG_SELFATTN = 0, 1       # whose out-fragment order the packed weight's k-slots follow


def pack_to_out_weight(weight, heads, kind=FRAG_XATTN):
    """to_out.weight [C, C] -> the fragment image sta_to_out_ln_ofrag streams through LDS (uint8 tensor); once per model.
    `kind`: the producer of the activations it will multiply — the cross-attention kernel (attn2.to_out) or the self-attention
    kernel (attn1.to_out); their out-fragment orders differ."""
    C = weight.shape[0]
    L = lib.load()
    n = L.sta_to_out_ln_packed_wo_bytes(C, heads)
    if n == 0 or weight.shape != (C, C) or not weight.is_cuda:
        raise ValueError("to_out + LayerNorm in out-fragment order: a square CUDA weight with C = 320, 8 heads; got %s heads=%d" % (tuple(weight.shape), heads))
    w = weight.detach().contiguous()
    buf = torch.empty(n, dtype=torch.uint8, device=w.device)
    lib.check(L.sta_to_out_ln_pack_wo(w.data_ptr(), buf.data_ptr(), C, heads, int(kind), _DT[w.dtype], _stream()), "sta_to_out_ln_pack_wo")
    return buf


def to_out_add_layernorm_ofrag(x, blended_ofrag, wo_packed, bias, ln_weight, ln_bias, eps, heads=8, y_qfrag=False):
    """s = x + blended . W_o^T + bias; returns (s, LayerNorm(s)) — `blended_ofrag` in the out-fragment order of the kernel that
    produced it (sta.ops.xattn_forward_proj(..., ofrag=True) or sta.ops.self_attention(..., sfrag=True); `wo_packed` packed for
    that kind); to_out's result never exists in HBM (csrc/sta_rowgemm.hip). `y_qfrag`: LayerNorm(s) comes back in query-fragment
    order (as add_layernorm(..., qfrag=True))."""
    C = x.shape[-1]
    R = x.numel() // C
    x = x.contiguous()
    s, y = torch.empty_like(x), torch.empty_like(x)
    lib.check(lib.load().sta_to_out_ln_ofrag(blended_ofrag.data_ptr(), wo_packed.data_ptr(), _ptr(bias), x.data_ptr(), ln_weight.data_ptr(),
                                             ln_bias.data_ptr(), s.data_ptr(), y.data_ptr(), R, C, heads, float(eps), int(bool(y_qfrag)), _DT[x.dtype], _stream()),
              "sta_to_out_ln_ofrag")
    return s, y


# Rows from which the persistent row-GEMM passes (to_out + LayerNorm: 128 rows per workgroup pass; GEGLU projection: 256) fill the
# chip: below it the library GEMM + the separate pass are taken (8 prompts per UNet call at 512^2 = 65536 rows at level 0).
ROWGEMM_MIN_ROWS = 65536


def _version(t):
    """t._version, or None for tensors without a version counter (tensors created under torch.inference_mode() raise on the
    attribute): without one an in-place update cannot be noticed, so nothing that depends on it is cached or trusted."""
    try:
        return t._version
    except RuntimeError:
        return None


def _producer_stats(x):
    """The per-channel sums the producing kernel left on `x` — only while x still holds what that kernel wrote (an in-place update
    since then bumps the tensor's version and the sums are ignored: the consumer then takes its own statistics pass)."""
    st = getattr(x, "_sta_stats", None)
    v = _version(x)
    return st[1] if st is not None and v is not None and st[0] == v else None


ROWGEMM_MAX_BYTES = 0xfffffff0 - 1     # what one launch of these passes addresses per tensor (32-bit buffer offsets)


def rows_addressable(x, width=None):
    """The row passes of csrc/sta_rowgemm.hip / sta_ffgemm.hip address every [rows, width] tensor of a launch through one 32-bit
    buffer descriptor and refuse rows * width * itemsize >= 4 GiB (`width`: the widest tensor of the chain, default x's own — the
    GEGLU output is [rows, 1280] at C = 320: 205 prompts per step at 512^2, 52 at 1024^2). The gate belongs in FRONT of the chain:
    once a producer has written fragment order there is no row-major fallback for its consumer."""
    rows = x.numel() // x.shape[-1]
    return rows * max(x.shape[-1], width or 0) * x.element_size() <= ROWGEMM_MAX_BYTES


def rowgemm_worthwhile(x, width=None):
    return x.numel() // x.shape[-1] >= ROWGEMM_MIN_ROWS and rows_addressable(x, width)


def pack_geglu_weight(weight):
    """GEGLU proj.weight [2 * inner, C] -> the fragment image sta_ff_geglu_qfrag streams through LDS (uint8 tensor); once per model."""
    two_inner, C = weight.shape
    L = lib.load()
    n = L.sta_ff_geglu_packed_w_bytes(C, two_inner // 2)
    if n == 0 or not weight.is_cuda:
        raise ValueError("fused GEGLU projection: a CUDA weight [2560, 320]; got %s" % (tuple(weight.shape),))
    w = weight.detach().contiguous()
    buf = torch.empty(n, dtype=torch.uint8, device=w.device)
    lib.check(L.sta_ff_geglu_pack_w(w.data_ptr(), buf.data_ptr(), C, two_inner // 2, _DT[w.dtype], _stream()), "sta_ff_geglu_pack_w")
    return buf


def ff_geglu_supported(C, inner):
    return bool(lib.load().sta_ff_geglu_packed_w_bytes(C, inner))


def ff_geglu_qfrag(y_qfrag, w_packed, bias, inner, h_frag=False):
    """h = (y W_v^T + b_v) * gelu(y W_g^T + b_g) with y in query-fragment order ([.., C] container); returns h [.., inner], row-major
    or (`h_frag`) in the fragment order ff_out_res_hfrag reads. The [.., 2 * inner] projection never exists in HBM (csrc/sta_ffgemm.hip)."""
    C = y_qfrag.shape[-1]
    R = y_qfrag.numel() // C
    h = torch.empty(*y_qfrag.shape[:-1], inner, dtype=y_qfrag.dtype, device=y_qfrag.device)
    lib.check(lib.load().sta_ff_geglu_qfrag(y_qfrag.data_ptr(), w_packed.data_ptr(), _ptr(bias), h.data_ptr(), R, C, inner, int(bool(h_frag)),
                                            _DT[y_qfrag.dtype], _stream()), "sta_ff_geglu_qfrag")
    return h


def pack_ff_out_weight(weight):
    """FeedForward net[2].weight [C, inner] -> the fragment image sta_ff_out_res_hfrag streams through LDS; once per model."""
    C, inner = weight.shape
    L = lib.load()
    n = L.sta_ff_out_packed_w_bytes(C, inner)
    if n == 0 or not weight.is_cuda:
        raise ValueError("fused feed-forward output: a CUDA weight [320, 1280]; got %s" % (tuple(weight.shape),))
    w = weight.detach().contiguous()
    buf = torch.empty(n, dtype=torch.uint8, device=w.device)
    lib.check(L.sta_ff_out_pack_w(w.data_ptr(), buf.data_ptr(), C, inner, _DT[w.dtype], _stream()), "sta_ff_out_pack_w")
    return buf


def ff_out_res_hfrag(x, h_frag, w_packed, bias):
    """x + h W2^T + b2 with h in fragment order (ff_geglu_qfrag(..., h_frag=True)); one pass (csrc/sta_ffgemm.hip)."""
    C, inner = x.shape[-1], h_frag.shape[-1]
    R = x.numel() // C
    x = x.contiguous()
    out = torch.empty_like(x)
    lib.check(lib.load().sta_ff_out_res_hfrag(h_frag.data_ptr(), w_packed.data_ptr(), _ptr(bias), x.data_ptr(), out.data_ptr(), R, C, inner,
                                              _DT[x.dtype], _stream()), "sta_ff_out_res_hfrag")
    return out


GN_STATS_FROM_PRODUCER = True   # convolutions / row GEMMs accumulate the consumer GroupNorm's statistics in their epilogue (no statistics pass)
CONV_MIN_ITEMS = 64     # (tile, channel part) work items below which the library convolution is the faster one (a launch feeds 256 CUs:
                        # at one prompt per step the 16 x 16 / 8 x 8 levels have 16 items and lose 2 - 3 x: tools/conv_bench.py --batch 2)
CONV3X3 = True          # the HIP implicit-GEMM 3x3 convolution for the NHWC trunk (csrc/sta_conv.hip); False: library convolution
_conv_zeros = {}
CONV_MAX_BYTES = 0xfffffff0 - 1     # output bytes one launch addresses (32-bit buffer descriptor)


def _finalize_stats(part, B, slots, C):
    """[B + 1, slots, C, 2] partial sums of a convolution / row GEMM epilogue -> [B, C, 2] (fixed order, no atomics)."""
    st = torch.empty((B, C, 2), dtype=torch.float32, device=part.device)
    lib.check(lib.load().sta_stats_finalize(part.data_ptr(), st.data_ptr(), B, slots, C, _stream()), "sta_stats_finalize")
    return st


def conv3x3_supported(x, weight, up2=False):
    """The HIP 3x3 convolution applies to NHWC 16-bit CUDA activations outside autograd at the geometries
    sta_conv3x3_nhwc_supported lists (the UNet's ResBlock / Upsample convolutions at 512^2 down to the 16x16 level, the VAE
    decoder's 128 / 256 / 512-channel convolutions); one image must stay below the 4 GiB a launch addresses."""
    if not (CONV3X3 and usable(x) and x.dim() == 4 and is_nhwc(x) and weight.dtype == x.dtype and tuple(weight.shape[2:]) == (3, 3)):
        return False
    B, Cin, H, W = x.shape
    if up2:
        H, W = 2 * H, 2 * W
    return bool(lib.load().sta_conv3x3_nhwc_supported(1, H, W, Cin, weight.shape[0])) and conv3x3_work_items(B, H, W, weight.shape[0]) >= CONV_MIN_ITEMS


def conv3x3_work_items(B, H, W, Cout):
    """(pixel tile, output-channel part) items of one sta_conv3x3_nhwc launch: what its 256 persistent workgroups share."""
    slots = lib.load().sta_conv3x3_stats_slots(H, W)
    tiles = (B + 1) // 2 if (H, W) == (8, 8) else B * slots // 4
    return tiles * (Cout // 160 if Cout % 160 == 0 else Cout // 128)


def pack_conv3x3_weight(weight):
    """Conv2d weight [Cout, Cin, 3, 3] (any memory format) -> the A-operand fragment image of sta_conv3x3_nhwc; once per model."""
    Cout, Cin = weight.shape[0], weight.shape[1]
    L = lib.load()
    n = L.sta_conv3x3_packed_w_bytes(Cin, Cout)
    if n == 0 or not weight.is_cuda or tuple(weight.shape[2:]) != (3, 3):
        raise ValueError("conv3x3: a CUDA weight [Cout %% 160 == 0 or Cout %% 128 == 0, Cin %% 64 == 0, 3, 3]; got %s" % (tuple(weight.shape),))
    w = weight.detach()
    buf = torch.empty(n, dtype=torch.uint8, device=w.device)
    so, si, sy, sx = w.stride()
    lib.check(L.sta_conv3x3_pack_w(w.data_ptr(), so, si, sy, sx, buf.data_ptr(), Cin, Cout, _DT[w.dtype], _stream()), "sta_conv3x3_pack_w")
    return buf


def conv3x3_nhwc(x, w_packed, Cout, up2=False, bias=None, res=None, stats=False):
    """conv2d(x, w, bias, padding=1) + res on an NHWC activation (logical shape [B, Cin, H, W], channels_last strides);
    up2: of the nearest-neighbour 2x upsampling of x, which is never written. Returns [B, Cout, H', W'] channels_last.
    A launch addresses its output through one 32-bit buffer descriptor: batches whose output exceeds 4 GiB go in several launches.
    stats: the kernel also accumulates every output channel's sum / sum of squares; they ride on the result as `_sta_stats`
    ([B, Cout, 2] fp32) for the GroupNorm that consumes it (groupnorm_silu skips its statistics pass)."""
    B, Cin, Hs, Ws = x.shape
    H, W = (2 * Hs, 2 * Ws) if up2 else (Hs, Ws)
    z = _zeros_page(x.device, 2 * Cin)
    out = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if res is not None:
        assert res.shape == out.shape and res.dtype == x.dtype and (is_nhwc(res) or res.is_contiguous(memory_format=torch.channels_last))
    if bias is not None:
        bias = bias.to(x.dtype).contiguous()
    esz = x.element_size()
    per_img = H * W * Cout * esz
    nb = max(1, min(B, CONV_MAX_BYTES // per_img))
    L = lib.load()
    slots = L.sta_conv3x3_stats_slots(H, W) if stats else 0
    part = torch.empty((B + 1, slots, Cout, 2), dtype=torch.float32, device=x.device) if slots else None
    for b0 in range(0, B, nb):
        n = min(nb, B - b0)
        # (a chunk's spare image slot is the next chunk's first image, written later; the last chunk's is the tensor's spare slot)
        lib.check(L.sta_conv3x3_nhwc(x.data_ptr() + b0 * Hs * Ws * Cin * esz, w_packed.data_ptr(), z.data_ptr(), _ptr(bias),
                                     0 if res is None else res.data_ptr() + b0 * per_img, out.data_ptr() + b0 * per_img,
                                     0 if part is None else part.data_ptr() + b0 * slots * Cout * 8, n, H, W, Cin, Cout,
                                     int(bool(up2)), _DT[x.dtype], _stream()), "sta_conv3x3_nhwc")
    if part is not None:
        if _version(out) is not None:           # (inference tensors: no producer statistics, the consumer takes its own pass)
            out._sta_stats = (out._version, _finalize_stats(part, B, slots, Cout))
    return out


def packed_conv_weight(owner, conv):
    """conv.weight as sta_conv3x3_nhwc streams it, cached on the owning module and repacked only when the weight tensor changes."""
    w = conv.weight
    key = (w.data_ptr(), _version(w), w.dtype)
    cache = owner.__dict__.setdefault("_sta_conv_cache", {})
    hit = cache.get(id(conv))
    if hit is None or hit[0] != key:
        hit = cache[id(conv)] = (key, pack_conv3x3_weight(w))
    return hit[1]


def conv3x3_module(owner, conv, x, bias=None, res=None, up2=False, stats=True):
    """conv(x) (+ bias + res) for a 3x3 nn.Conv2d outside autograd: the HIP convolution where it applies, the library convolution
    (+ the fused bias / residual pass) elsewhere. `bias=None` means bias-free (the caller folds conv.bias into a later pass).
    `stats=False`: the result is not read by a GroupNorm next (e.g. it feeds an Upsample convolution) — no epilogue statistics, no
    partial buffer, no finalize launch."""
    if conv3x3_supported(x, conv.weight, up2=up2) and (res is None or is_nhwc(res)):
        return conv3x3_nhwc(x, packed_conv_weight(owner, conv), conv.weight.shape[0], up2=up2, bias=bias, res=res,
                            stats=GN_STATS_FROM_PRODUCER and stats)
    if up2:
        x = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
    h = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding)
    return h if bias is None and res is None else add_bias_nchw(h if res is None else res, None if res is None else h, bias)


CAT_IN_PLACE = True     # output blocks read `cat([h, skip])` in place (GroupNorm and the 1x1 skip GEMM take both tensors); False: torch.cat


def groupnorm_silu_cat(xa, xb, weight, bias, groups, eps, silu=True):
    """act(GroupNorm(cat([xa, xb], dim=1))) for two NHWC activations, the concatenation read in place (csrc/sta_unet.hip);
    returns the normalised concatenated tensor. Callers check cat_supported first."""
    B, Ca = xa.shape[0], xa.shape[1]
    C = Ca + xb.shape[1]
    HW = xa.numel() // (B * Ca)
    L = lib.load()
    y = torch.empty((B, C) + tuple(xa.shape[2:]), dtype=xa.dtype, device=xa.device, memory_format=torch.channels_last)
    sa, sb = (_producer_stats(xa), _producer_stats(xb)) if GN_STATS_FROM_PRODUCER else (None, None)
    if sa is not None and sb is not None:
        lib.check(L.sta_groupnorm_silu_nhwc_cstats(xa.data_ptr(), xb.data_ptr(), Ca, sa.data_ptr(), sb.data_ptr(), None, weight.data_ptr(),
                                                   bias.data_ptr(), y.data_ptr(), B, C, HW, groups, float(eps), int(bool(silu)), _DT[xa.dtype],
                                                   _stream()), "sta_groupnorm_silu_nhwc_cstats")
        return y
    ws = torch.empty(L.sta_groupnorm_nhwc_workspace_bytes(B, HW, groups) // 4, dtype=torch.float32, device=xa.device)
    lib.check(L.sta_groupnorm_silu_nhwc_cat(xa.data_ptr(), xb.data_ptr(), Ca, None, weight.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                            ws.data_ptr(), B, C, HW, groups, float(eps), int(bool(silu)), _DT[xa.dtype], _stream()),
              "sta_groupnorm_silu_nhwc_cat")
    return y


def cat_supported(xa, xb, groups, skip_weight):
    """Both halves NHWC 16-bit outside autograd, channel counts the in-place GroupNorm and the in-place 1x1 skip GEMM take."""
    if not (CAT_IN_PLACE and usable(xa) and is_nhwc(xa) and is_nhwc(xb) and xa.dtype == xb.dtype and xa.shape[0] == xb.shape[0]
            and xa.shape[2:] == xb.shape[2:]):
        return False
    Ca, C = xa.shape[1], xa.shape[1] + xb.shape[1]
    if Ca % 64 or C % 8 or C % groups or (C // groups < 8 and C // groups != 4) or C > 4096 or groups > 64:
        return False
    rows = xa.numel() // Ca
    return (LINEAR_ROWS and rows >= LINEAR_MIN_ROWS and skip_weight.dtype == xa.dtype
            and bool(lib.load().sta_linear_rows_supported(rows, C, skip_weight.shape[0])))


def linear_rows_cat(xa, xb, w_packed, N, bias=None, res=None):
    """cat([xa, xb], dim=-1) @ W^T + bias + res over row tensors [..., Ka] and [..., Kb], the concatenation read in place."""
    Ka, Kb = xa.shape[-1], xb.shape[-1]
    R = xa.numel() // Ka
    assert xa.is_contiguous() and xb.is_contiguous() and xb.numel() // Kb == R and Ka % 64 == 0
    out = torch.empty(xa.shape[:-1] + (N,), dtype=xa.dtype, device=xa.device)
    if bias is not None:
        bias = bias.to(xa.dtype).contiguous()
    z = _zeros_page(xa.device, 2 * (Ka + Kb))
    lib.check(lib.load().sta_linear_rows_cat(xa.data_ptr(), xb.data_ptr(), Ka, w_packed.data_ptr(), z.data_ptr(), _ptr(bias), _ptr(res),
                                             out.data_ptr(), R, Ka + Kb, N, _DT[xa.dtype], _stream()), "sta_linear_rows_cat")
    return out


LINEAR_ROWS = True      # the HIP row GEMM for Linear layers / 1x1 convolutions outside autograd (csrc/sta_gemm.hip); False: library GEMM
LINEAR_MIN_ROWS = 4096  # below this many rows the library GEMM stays (time-embedding, context projections)


_retired_pages = []


def _zeros_page(device, nbytes):
    """The page of zeros the convolution / row GEMM kernels read their padding from. Captured hipGraphs hold its address: a page
    that had to grow is kept alive, never freed."""
    z = _conv_zeros.get(device)
    if z is None or z.numel() < nbytes:
        if z is not None:
            _retired_pages.append(z)
        z = _conv_zeros[device] = torch.zeros(max(nbytes, 16384), dtype=torch.uint8, device=device)
    return z


def linear_rows_supported(x, weight):
    """The HIP row GEMM applies to dense 16-bit CUDA rows outside autograd: x [..., K] contiguous, weight [N, K] (or [N, K, 1, 1])
    with K % 64 == 0 and N % 160 == 0 or N % 128 == 0, at least LINEAR_MIN_ROWS rows, output below 4 GiB."""
    if not (LINEAR_ROWS and x.is_cuda and x.dtype in _DT and not torch.is_grad_enabled() and ENABLED and not _hold and weight.dtype == x.dtype):
        return False
    K = x.shape[-1]
    R = x.numel() // K
    return R >= LINEAR_MIN_ROWS and x.is_contiguous() and bool(lib.load().sta_linear_rows_supported(R, K, weight.shape[0]))


def pack_linear_weight(weight):
    """Linear weight [N, K] (or a 1x1 Conv2d weight [N, K, 1, 1], any memory format) -> the A-operand fragment image of sta_linear_rows."""
    N, K = weight.shape[0], weight.shape[1]
    L = lib.load()
    n = L.sta_linear_rows_packed_w_bytes(K, N)
    if n == 0 or not weight.is_cuda:
        raise ValueError("linear_rows: a CUDA weight [N %% 160 == 0 or N %% 128 == 0, K %% 64 == 0]; got %s" % (tuple(weight.shape),))
    w = weight.detach()
    buf = torch.empty(n, dtype=torch.uint8, device=w.device)
    lib.check(L.sta_linear_rows_pack_w(w.data_ptr(), w.stride(0), w.stride(1), buf.data_ptr(), K, N, _DT[w.dtype], _stream()), "sta_linear_rows_pack_w")
    return buf


def linear_rows(x, w_packed, N, bias=None, res=None, stats_rows=None):
    """x @ W^T + bias + res over the rows of a contiguous [..., K] tensor; returns [..., N] (csrc/sta_gemm.hip).
    stats_rows = rows per image (a multiple of 256): per-image, per-column sum / sum of squares ride on the result as `_sta_stats`."""
    K = x.shape[-1]
    R = x.numel() // K
    out = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
    if res is not None:
        assert res.numel() == out.numel() and res.dtype == x.dtype and res.is_contiguous()
    if bias is not None:
        bias = bias.to(x.dtype).contiguous()
    z = _zeros_page(x.device, 2 * K)
    if stats_rows and stats_rows % 256 == 0 and R % stats_rows == 0:
        n_img, slots = R // stats_rows, stats_rows // 64
        part = torch.empty((n_img + 1, slots, N, 2), dtype=torch.float32, device=x.device)
        lib.check(lib.load().sta_linear_rows_stats(x.data_ptr(), w_packed.data_ptr(), z.data_ptr(), _ptr(bias), _ptr(res), out.data_ptr(),
                                                   part.data_ptr(), stats_rows, R, K, N, _DT[x.dtype], _stream()), "sta_linear_rows_stats")
        if _version(out) is not None:
            out._sta_stats = (out._version, _finalize_stats(part, n_img, slots, N))
        return out
    lib.check(lib.load().sta_linear_rows(x.data_ptr(), w_packed.data_ptr(), z.data_ptr(), _ptr(bias), _ptr(res), out.data_ptr(), R, K, N,
                                         _DT[x.dtype], _stream()), "sta_linear_rows")
    return out


def packed_linear_weight(owner, key_obj, weight):
    """weight as sta_linear_rows streams it, cached on the owning module and repacked only when the weight tensor changes."""
    key = (weight.data_ptr(), _version(weight), weight.dtype)
    cache = owner.__dict__.setdefault("_sta_linear_cache", {})
    hit = cache.get(id(key_obj))
    if hit is None or hit[0] != key:
        hit = cache[id(key_obj)] = (key, pack_linear_weight(weight))
    return hit[1]


def linear_module(lin, x, res=None):
    """lin(x) (+ res) for an nn.Linear outside autograd: the HIP row GEMM where it applies, the library GEMM elsewhere."""
    if linear_rows_supported(x, lin.weight) and (res is None or res.is_contiguous()):
        return linear_rows(x, packed_linear_weight(lin, lin, lin.weight), lin.weight.shape[0], bias=lin.bias, res=res)
    y = torch.nn.functional.linear(x, lin.weight, lin.bias)
    return y if res is None else y + res


def add_bias_nchw(a, b=None, bias=None):
    """a + b + bias[None, :, None, None] over 4-D activations (both NCHW-contiguous or both channels_last)."""
    B, C = a.shape[0], a.shape[1]
    HW = a.numel() // (B * C)
    if (C % 8 if is_nhwc(a) else HW % 8):
        t = a if b is None else a + b
        return t if bias is None else t + bias[None, :, None, None]
    y = torch.empty_like(a)
    if is_nhwc(a):
        if b is not None and not is_nhwc(b):
            b = b.contiguous(memory_format=torch.channels_last)
        lib.check(lib.load().sta_add_bias_rows(a.data_ptr(), _ptr(b), _ptr(bias), y.data_ptr(), B * HW, C, _DT[a.dtype], _stream()),
                  "sta_add_bias_rows")
        return y
    if b is not None and not b.is_contiguous():
        b = b.contiguous()
    lib.check(lib.load().sta_add_bias_nchw(a.data_ptr(), _ptr(b), _ptr(bias), y.data_ptr(), B, C, HW, _DT[a.dtype], _stream()),
              "sta_add_bias_nchw")
    return y


# -- differentiable forms for the tracked epochs ----------------------------------------------------------------------
def tracked_usable(x):
    """The differentiable fused glue ops apply while autograd records, on dense 16-bit CUDA activations (4-D ones in NHWC)."""
    return (TRACKED and ENABLED and x.is_cuda and x.dtype in _DT and torch.is_grad_enabled()
            and (is_nhwc(x) if x.dim() == 4 else x.is_contiguous()))


class GroupNormSiLUFn(torch.autograd.Function):
    """act(GroupNorm(x + add[:, :, None, None])) on an NHWC activation; backward = sta_groupnorm_silu_nhwc_bwd."""

    @staticmethod
    def forward(ctx, x, add, weight, bias, groups, eps, silu):
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        L = lib.load()
        add = None if add is None else add.float().contiguous()
        y = torch.empty_like(x)
        ws = torch.empty(L.sta_groupnorm_nhwc_workspace_bytes(B, HW, groups) // 4, dtype=torch.float32, device=x.device)
        lib.check(L.sta_groupnorm_silu_nhwc(x.data_ptr(), _ptr(add), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), ws.data_ptr(),
                                            B, C, HW, groups, float(eps), int(bool(silu)), _DT[x.dtype], _stream()), "sta_groupnorm_silu_nhwc")
        ctx.save_for_backward(x, add, weight, bias, ws)
        ctx.cfg = (B, C, HW, groups, float(eps), int(bool(silu)))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, add, weight, bias, ws = ctx.saved_tensors
        B, C, HW, groups, eps, silu = ctx.cfg
        if not is_nhwc(dy):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        wsb = torch.empty_like(ws)
        lib.check(lib.load().sta_groupnorm_silu_nhwc_bwd(x.data_ptr(), _ptr(add), weight.data_ptr(), bias.data_ptr(), dy.data_ptr(),
                                                         dx.data_ptr(), ws.data_ptr(), wsb.data_ptr(), B, C, HW, groups, eps, silu,
                                                         _DT[x.dtype], _stream()), "sta_groupnorm_silu_nhwc_bwd")
        return dx, None, None, None, None, None, None


def groupnorm_silu_tracked(x, weight, bias, groups, eps, add=None, silu=True):
    """Differentiable GroupNorm(+pre-add)(+SiLU) for NHWC activations (eager chain outside the kernel's limits)."""
    C = x.shape[1]
    if add is not None and add.requires_grad:
        raise RuntimeError("the pre-add of the fused GroupNorm carries no gradient (timestep embedding)")
    if not is_nhwc(x) or C % 8 or (C // groups < 8 and C // groups != 4) or C > 4096 or groups > 64:
        h = x if add is None else x + add.to(x.dtype)[:, :, None, None]
        h = torch.nn.functional.group_norm(h, groups, weight, bias, eps)
        return torch.nn.functional.silu(h) if silu else h
    return GroupNormSiLUFn.apply(x, add, weight, bias, groups, eps, silu)


class GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return geglu(h)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        D = h.shape[-1] // 2
        dy = dy.contiguous()
        dh = torch.empty_like(h)
        lib.check(lib.load().sta_geglu_bwd(h.data_ptr(), dy.data_ptr(), dh.data_ptr(), h.numel() // (2 * D), D, _DT[h.dtype], _stream()),
                  "sta_geglu_bwd")
        return dh


def geglu_tracked(h):
    if (h.shape[-1] // 2) % 8 or not h.is_contiguous():
        a, gate = h.chunk(2, dim=-1)
        return a * torch.nn.functional.gelu(gate)
    return GegluFn.apply(h)


class AddLayerNormFn(torch.autograd.Function):
    """(s, y) = (x + f + bias, LayerNorm(s)); backward: ds_total = ds + dLayerNorm(dy), handed to x and f alike."""

    @staticmethod
    def forward(ctx, x, f, bias, ln_weight, ln_bias, eps):
        s, y = add_layernorm(x, f, bias, ln_weight, ln_bias, eps, store_sum=True)
        ctx.save_for_backward(s, ln_weight)
        ctx.eps, ctx.has_f = float(eps), f is not None
        return s, y

    @staticmethod
    def backward(ctx, ds, dy):
        s, ln_weight = ctx.saved_tensors
        C = s.shape[-1]
        dy = dy.contiguous()
        ds = None if ds is None else ds.contiguous()
        g = torch.empty_like(s)
        lib.check(lib.load().sta_layernorm_bwd(s.data_ptr(), ln_weight.data_ptr(), dy.data_ptr(), _ptr(ds), g.data_ptr(), s.numel() // C, C,
                                               ctx.eps, _DT[s.dtype], _stream()), "sta_layernorm_bwd")
        return g, (g if ctx.has_f else None), None, None, None, None


def add_layernorm_tracked(x, f, bias, ln_weight, ln_bias, eps):
    """Differentiable (x + f + bias, LayerNorm(x + f + bias)); x, f [.., C] contiguous."""
    C = x.shape[-1]
    if C % 8 or C > 2048 or not x.is_contiguous() or (f is not None and not f.is_contiguous()):
        t = x if f is None else x + f
        t = t if bias is None else t + bias
        return t, torch.nn.functional.layer_norm(t, (C,), ln_weight, ln_bias, eps)
    return AddLayerNormFn.apply(x, f, bias, ln_weight, ln_bias, eps)


class AddBiasFn(torch.autograd.Function):
    """a + b + bias[None, :, None, None] (NHWC); backward hands dy to a and b."""

    @staticmethod
    def forward(ctx, a, b, bias):
        ctx.has_b = b is not None
        return add_bias_nchw(a, b, bias)

    @staticmethod
    def backward(ctx, dy):
        return dy, (dy if ctx.has_b else None), None


class Conv3x3Fn(torch.autograd.Function):
    """conv2d(x, w, padding=1) without bias on an NHWC activation with FROZEN weights: forward and input gradient are both
    csrc/sta_conv.hip — the input gradient of a 3x3 / stride-1 / padding-1 convolution is the same convolution with the weight's
    channel axes exchanged and its taps mirrored (packed once per model next to the forward image)."""

    @staticmethod
    def forward(ctx, x, wp_fwd, wp_bwd, cin, cout):
        ctx.wp_bwd, ctx.cin = wp_bwd, cin
        return conv3x3_nhwc(x, wp_fwd, cout)

    @staticmethod
    def backward(ctx, dy):
        if not is_nhwc(dy):
            dy = dy.contiguous(memory_format=torch.channels_last)
        return conv3x3_nhwc(dy, ctx.wp_bwd, ctx.cin), None, None, None, None


def conv3x3_tracked_supported(x, weight):
    """The differentiable HIP convolution applies while autograd records (tracked epochs, fused.TRACKED) to NHWC 16-bit activations
    and frozen 3x3 weights whose forward AND mirrored images both fit the kernel's channel rules."""
    if not (CONV3X3 and tracked_usable(x) and x.dim() == 4 and is_nhwc(x) and weight.dtype == x.dtype and not weight.requires_grad
            and tuple(weight.shape[2:]) == (3, 3)):
        return False
    B, Cin, H, W = x.shape
    L = lib.load()
    Cout = weight.shape[0]
    return (bool(L.sta_conv3x3_nhwc_supported(1, H, W, Cin, Cout) and L.sta_conv3x3_nhwc_supported(1, H, W, Cout, Cin))
            and min(conv3x3_work_items(B, H, W, Cout), conv3x3_work_items(B, H, W, Cin)) >= CONV_MIN_ITEMS)


def conv3x3_tracked(owner, conv, x):
    """conv(x) WITHOUT its bias for a frozen 3x3 nn.Conv2d under autograd (callers fold the bias into the pass that follows)."""
    w = conv.weight
    key = (w.data_ptr(), _version(w), w.dtype)
    cache = owner.__dict__.setdefault("_sta_conv_bwd_cache", {})
    hit = cache.get(id(conv))
    if hit is None or hit[0] != key:
        with torch.no_grad():
            hit = cache[id(conv)] = (key, pack_conv3x3_weight(w.detach().permute(1, 0, 2, 3).flip(2, 3)))
    return Conv3x3Fn.apply(x, packed_conv_weight(owner, conv), hit[1], w.shape[1], w.shape[0])


def add_bias_tracked(a, b=None, bias=None):
    if not is_nhwc(a) or a.shape[1] % 8 or (b is not None and not is_nhwc(b)):
        t = a if b is None else a + b
        return t if bias is None else t + bias[None, :, None, None]
    return AddBiasFn.apply(a, b, bias)
