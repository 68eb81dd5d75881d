"""GPU parity of the trunk glue kernels (include/sta_unet.h) against plain PyTorch fp32 references of the same
chains (floating-point kernels: tolerance = a few ulps of the 16-bit output, stated per test)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _kernels_at_every_size(monkeypatch):
    """The product routes launches with few work items to the library (sta.fused.CONV_MIN_ITEMS); the kernel tests cover small shapes too."""
    from sta import fused
    monkeypatch.setattr(fused, "CONV_MIN_ITEMS", 1)

EPS = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


def _close(got, ref, dtype, k=3.0):
    err = (got.float().cpu() - ref).abs()
    tol = k * EPS[dtype] * (1.0 + ref.abs())
    assert (err <= tol).all(), "max err %.4g at tol %.4g" % (err.max().item(), tol.max().item())


@pytest.mark.parametrize("B,C,H,G", [
    (2, 320, 64, 32),     # level 0, slab in registers (40960 elements)
    (2, 960, 64, 32),     # concat input of out.9: 122880-element slab, two-pass path
    (4, 1920, 32, 32),    # level 1 concat
    (2, 1280, 8, 32),     # level 3: 2560-element slabs (256-thread variant)
    (2, 2560, 16, 32),    # level 2 concat
    (3, 64, 12, 8),       # HW = 144 (768^2 mid level), odd batch
    (2, 320, 96, 32),     # 768^2 level 0: 92160-element slab
])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_add,silu", [(False, True), (True, True), (False, False)])
def test_groupnorm_silu(B, C, H, G, dtype, with_add, silu):
    from sta import fused
    g = torch.Generator().manual_seed(B * C + H)
    x = (torch.randn(B, C, H, H, generator=g) * 1.7 + 0.3).to(dtype)
    w = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dtype)
    b = (0.2 * torch.randn(C, generator=g)).to(dtype)
    add = torch.randn(B, C, generator=g) if with_add else None
    xin = x.float() + (add[:, :, None, None] if with_add else 0.0)
    ref = F.group_norm(xin, G, w.float(), b.float(), 1e-5)
    ref = F.silu(ref) if silu else ref
    got = fused.groupnorm_silu(x.cuda(), w.cuda(), b.cuda(), G, 1e-5, add=None if add is None else add.cuda(), silu=silu)
    torch.cuda.synchronize()
    _close(got, ref, dtype)


@pytest.mark.parametrize("R,D", [(4096, 1280), (1000, 2560), (64, 5120), (7, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_geglu(R, D, dtype):
    from sta import fused
    g = torch.Generator().manual_seed(R + D)
    h = (torch.randn(R, 2 * D, generator=g) * 2).to(dtype)
    a, gate = h.float().chunk(2, dim=-1)
    got = fused.geglu(h.cuda())
    torch.cuda.synchronize()
    _close(got, a * F.gelu(gate), dtype)


@pytest.mark.parametrize("R,C", [(8192, 320), (2048, 640), (513, 1280), (3, 2048), (5, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_f,with_bias", [(True, True), (True, False), (False, False), (False, True)])
def test_add_layernorm(R, C, dtype, with_f, with_bias):
    from sta import fused
    g = torch.Generator().manual_seed(R + C)
    x = torch.randn(R, C, generator=g).to(dtype)
    f = (torch.randn(R, C, generator=g) * 0.5).to(dtype) if with_f else None
    bias = (torch.randn(C, generator=g) * 0.3).to(dtype) if with_bias else None
    w = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dtype)
    b = (0.2 * torch.randn(C, generator=g)).to(dtype)
    s_ref = x.float() + (f.float() if with_f else 0.0) + (bias.float() if with_bias else 0.0)
    s, y = fused.add_layernorm(x.cuda(), None if f is None else f.cuda(), None if bias is None else bias.cuda(), w.cuda(), b.cuda(), 1e-5)
    torch.cuda.synchronize()
    _close(s, s_ref, dtype, k=1.0)
    # the kernel normalises the ROUNDED sum (that is what the residual stream carries on)
    y_ref = F.layer_norm(s.float().cpu(), (C,), w.float(), b.float(), 1e-5)
    _close(y, y_ref, dtype)
    s2, y2 = fused.add_layernorm(x.cuda(), None if f is None else f.cuda(), None if bias is None else bias.cuda(), w.cuda(), b.cuda(), 1e-5,
                                 store_sum=False)
    assert s2 is None
    _close(y2, F.layer_norm(s_ref, (C,), w.float(), b.float(), 1e-5), dtype)


@pytest.mark.parametrize("R,C", [(8192, 320), (4096, 160), (16, 512), (48, 32), (2048, 640), (32, 1024), (64, 544)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_f,with_bias,store", [(True, True, True), (True, False, True), (False, False, False)])
def test_add_layernorm_query_fragment_order(R, C, dtype, with_f, with_bias, store):
    """sta_add_layernorm_qfrag = sta_add_layernorm with y laid out as the projection-fused attention kernel's MFMA B operands
    (16-row groups -> C/32 fragments of 1 KiB): bit-identical values at permuted addresses, s untouched; shapes outside its
    limits are refused by the C-ABI."""
    from sta import fused, lib, ops
    g = torch.Generator().manual_seed(R + C)
    x = torch.randn(R, C, generator=g).to(dtype).cuda()
    f = (torch.randn(R, C, generator=g) * 0.5).to(dtype).cuda() if with_f else None
    bias = (torch.randn(C, generator=g) * 0.3).to(dtype).cuda() if with_bias else None
    w = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dtype).cuda()
    b = (0.2 * torch.randn(C, generator=g)).to(dtype).cuda()
    s0, y0 = fused.add_layernorm(x, f, bias, w, b, 1e-5, store_sum=store)
    s1, y1 = fused.add_layernorm(x, f, bias, w, b, 1e-5, store_sum=store, qfrag=True)
    torch.cuda.synchronize()
    assert (s0 is None and s1 is None) or torch.equal(s0, s1)
    assert torch.equal(ops.from_qfrag(y1), y0) and torch.equal(ops.to_qfrag(y0), y1)
    # the layout, spelled out: group P, fragment s, lane 16 g + c holds y[16 P + c, 32 s + 8 g .. + 7]
    flat = y1.reshape(-1)
    for (P, sfr, gl, c) in [(0, 0, 0, 0), (R // 16 - 1, C // 32 - 1, 3, 15), (R // 32, 0, 2, 7)]:
        off = ((P * (C // 32) + sfr) * 64 + 16 * gl + c) * 8
        assert torch.equal(flat[off:off + 8], y0[16 * P + c, 32 * sfr + 8 * gl:32 * sfr + 8 * gl + 8])
    L = lib.load()
    for (r_bad, c_bad) in [(R + 8, C), (R, 16), (R, 1056)]:
        rc = L.sta_add_layernorm_qfrag(x.data_ptr(), 0, 0, w.data_ptr(), b.data_ptr(), 0, y1.data_ptr(), r_bad, c_bad, 1e-5, lib.STA_F16, 0)
        assert rc == -1 and "qfrag" in lib.last_error()


@pytest.mark.parametrize("R", [16, 128, 4096 + 48, 2 * 4096 * 3])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_bias", [True, False])
def test_to_out_add_layernorm_out_fragment_order(R, dtype, with_bias):
    """sta_to_out_ln_ofrag (csrc/sta_rowgemm.hip): s = x + blended . W^T + bias, y = LayerNorm(s), with `blended` in the out-fragment
    order of the head-pair attention kernel and W streamed through LDS, against fp64 torch on the same 16-bit inputs. Replaces
    to_out (attention.py:215) + the residual + norm3 (:294-299). Row counts: one item, one pass, a partial last workgroup pass,
    several passes per workgroup."""
    from sta import fused, lib, ops
    C, heads = 320, 8
    g = torch.Generator().manual_seed(R)
    bl = torch.randn(R, C, generator=g).to(dtype)
    x = torch.randn(R, C, generator=g).to(dtype)
    w = (torch.randn(C, C, generator=g) / C ** 0.5).to(dtype)
    bias = (torch.randn(C, generator=g) * 0.3).to(dtype) if with_bias else None
    lw = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dtype)
    lb = (0.2 * torch.randn(C, generator=g)).to(dtype)
    wo = fused.pack_to_out_weight(w.cuda(), heads)
    s, y = fused.to_out_add_layernorm_ofrag(x.cuda(), ops.to_ofrag(bl.cuda()), wo, None if bias is None else bias.cuda(), lw.cuda(), lb.cuda(), 1e-5, heads)
    torch.cuda.synchronize()
    s_ref = x.double() + bl.double() @ w.double().t() + (bias.double() if with_bias else 0.0)
    _close(s, s_ref.float(), dtype, k=1.0)
    # the kernel normalises the ROUNDED sum (what the residual stream carries on), as sta_add_layernorm does
    y_ref = F.layer_norm(s.float().cpu(), (C,), lw.float(), lb.float(), 1e-5)
    _close(y, y_ref, dtype)
    # and it is the same function as the unfused pair it replaces (library GEMM + sta_add_layernorm), up to the GEMM result's own rounding
    s2, y2 = fused.add_layernorm(x.cuda(), F.linear(bl.cuda(), w.cuda(), None if bias is None else bias.cuda()), None, lw.cuda(), lb.cuda(), 1e-5)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert ((s.float() - s2.float()).abs() <= 4 * eps * (1.0 + s2.float().abs())).all()
    # the self-attention side of the block: the activations in the SELF-attention kernel's out-fragment order, the weight packed for
    # it, and y = norm2(s) written in query-fragment order for the cross-attention kernel — the same values
    wo1 = fused.pack_to_out_weight(w.cuda(), heads, fused.FRAG_SELFATTN)
    s3, y3 = fused.to_out_add_layernorm_ofrag(x.cuda(), ops.to_sfrag(bl.cuda()), wo1, None if bias is None else bias.cuda(), lw.cuda(), lb.cuda(), 1e-5, heads,
                                              y_qfrag=True)
    torch.cuda.synchronize()
    # (the k-slots of the MFMAs hold the channels in another order: the fp32 sums may differ in the last bit, so the two are held to
    # the fp64 reference separately, not to each other)
    _close(s3, s_ref.float(), dtype, k=1.0)
    _close(ops.from_qfrag(y3), F.layer_norm(s3.float().cpu(), (C,), lw.float(), lb.float(), 1e-5), dtype)
    assert (s3 != s).float().mean() < 0.02
    L = lib.load()
    assert L.sta_to_out_ln_packed_wo_bytes(640, 8) == 0 and L.sta_to_out_ln_packed_wo_bytes(320, 4) == 0
    rc = L.sta_to_out_ln_ofrag(x.cuda().data_ptr(), wo.data_ptr(), 0, x.cuda().data_ptr(), lw.cuda().data_ptr(), lb.cuda().data_ptr(), s.data_ptr(), y.data_ptr(),
                               R + 8, C, heads, 1e-5, 0, lib.STA_F16, 0)
    assert rc == -1 and "multiple of 16" in lib.last_error()


@pytest.mark.parametrize("R", [16, 256, 4096 + 48, 256 * 9])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_bias", [True, False])
def test_ff_geglu_query_fragment_order(R, dtype, with_bias):
    """sta_ff_geglu_qfrag (csrc/sta_ffgemm.hip): h = (y Wv^T + bv) * gelu(y Wg^T + bg) from y in query-fragment order, the weight
    streamed through LDS, against fp64 torch on the same 16-bit inputs (exact-erf GELU). Replaces GEGLU.forward's projection GEMM +
    chunk + gelu + mul (attention.py:42-45). Row counts: one item, one pass, a ragged last pass, several passes per workgroup."""
    from sta import fused, lib, ops
    C, inner = 320, 1280
    g = torch.Generator().manual_seed(R + 7)
    y = torch.randn(R, C, generator=g).to(dtype)
    w = (torch.randn(2 * inner, C, generator=g) / C ** 0.5).to(dtype)
    bias = (torch.randn(2 * inner, generator=g) * 0.3).to(dtype) if with_bias else None
    wp = fused.pack_geglu_weight(w.cuda())
    h = fused.ff_geglu_qfrag(ops.to_qfrag(y.cuda()), wp, None if bias is None else bias.cuda(), inner)
    torch.cuda.synchronize()
    proj = y.double() @ w.double().t() + (bias.double() if with_bias else 0.0)
    ref = proj[:, :inner] * F.gelu(proj[:, inner:])
    _close(h, ref.float(), dtype, k=2.0)
    # and against the pair it replaces (library GEMM rounding its result to 16 bit, then sta_geglu)
    h2 = fused.geglu(F.linear(y.cuda(), w.cuda(), None if bias is None else bias.cuda()))
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert ((h.float() - h2.float()).abs() <= 16 * eps * (1.0 + h2.float().abs())).all()
    # h in fragment order + the second half of the feed-forward with the block's residual (sta_ff_out_res_hfrag): out = x + h W2^T + b2
    hf = fused.ff_geglu_qfrag(ops.to_qfrag(y.cuda()), wp, None if bias is None else bias.cuda(), inner, h_frag=True)
    assert torch.equal(ops.from_qfrag(hf), h)                       # the same values; h's fragment order is query-fragment order at C = inner
    x = torch.randn(R, C, generator=g).to(dtype)
    w2 = (torch.randn(C, inner, generator=g) / inner ** 0.5).to(dtype)
    b2 = (torch.randn(C, generator=g) * 0.3).to(dtype) if with_bias else None
    out = fused.ff_out_res_hfrag(x.cuda(), hf, fused.pack_ff_out_weight(w2.cuda()), None if b2 is None else b2.cuda())
    torch.cuda.synchronize()
    ref2 = x.double() + h.double().cpu() @ w2.double().t() + (b2.double() if with_bias else 0.0)
    _close(out, ref2.float(), dtype, k=1.0)
    L = lib.load()
    assert L.sta_ff_geglu_packed_w_bytes(640, 2560) == 0 and L.sta_ff_geglu_packed_w_bytes(320, 640) == 0 and L.sta_ff_out_packed_w_bytes(640, 2560) == 0
    rc = L.sta_ff_geglu_qfrag(h.data_ptr(), wp.data_ptr(), 0, h.data_ptr(), R + 8, C, inner, 0, lib.STA_F16, 0)
    assert rc == -1 and "multiple of 16" in lib.last_error()


@pytest.mark.parametrize("B,C,H", [(2, 320, 64), (3, 1280, 8), (2, 64, 12)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_add_bias_nchw(B, C, H, dtype):
    from sta import fused
    g = torch.Generator().manual_seed(C + H)
    a = torch.randn(B, C, H, H, generator=g).to(dtype)
    b = torch.randn(B, C, H, H, generator=g).to(dtype)
    bias = torch.randn(C, generator=g).to(dtype)
    got = fused.add_bias_nchw(a.cuda(), b.cuda(), bias.cuda())
    torch.cuda.synchronize()
    _close(got, a.float() + b.float() + bias.float()[None, :, None, None], dtype, k=1.0)
    _close(fused.add_bias_nchw(a.cuda(), None, bias.cuda()), a.float() + bias.float()[None, :, None, None], dtype, k=1.0)
    _close(fused.add_bias_nchw(a.cuda(), b.cuda(), None), a.float() + b.float(), dtype, k=1.0)


@pytest.mark.parametrize("B,C,H,G", [(2, 320, 64, 32), (2, 960, 64, 32), (4, 1920, 32, 32), (2, 1280, 8, 32), (2, 2560, 16, 32),
                                     (3, 256, 12, 32), (2, 640, 96, 32)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_add,silu", [(False, True), (True, True), (False, False)])
def test_groupnorm_silu_channels_last(B, C, H, G, dtype, with_add, silu):
    """NHWC variant (two kernels: per-chunk moments, then normalise): same reference, channels_last in and out."""
    from sta import fused
    g = torch.Generator().manual_seed(B * C + H)
    x = (torch.randn(B, C, H, H, generator=g) * 1.7 + 0.3).to(dtype)
    w = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dtype)
    b = (0.2 * torch.randn(C, generator=g)).to(dtype)
    add = torch.randn(B, C, generator=g) if with_add else None
    xin = x.float() + (add[:, :, None, None] if with_add else 0.0)
    ref = F.group_norm(xin, G, w.float(), b.float(), 1e-5)
    ref = F.silu(ref) if silu else ref
    xc = x.cuda().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert fused.is_nhwc(xc) and fused.usable(xc)
    got = fused.groupnorm_silu(xc, w.cuda(), b.cuda(), G, 1e-5, add=None if add is None else add.cuda(), silu=silu)
    torch.cuda.synchronize()
    assert fused.is_nhwc(got)
    _close(got, ref, dtype)


def test_add_bias_channels_last():
    from sta import fused
    g = torch.Generator().manual_seed(9)
    a = torch.randn(2, 320, 16, 16, generator=g).bfloat16()
    b = torch.randn(2, 320, 16, 16, generator=g).bfloat16()
    bias = torch.randn(320, generator=g).bfloat16()
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)
    got = fused.add_bias_nchw(cl(a), cl(b), bias.cuda())
    assert fused.is_nhwc(got)
    _close(got, a.float() + b.float() + bias.float()[None, :, None, None], torch.bfloat16, k=1.0)
    _close(fused.add_bias_nchw(cl(a), b.cuda(), None), a.float() + b.float(), torch.bfloat16, k=1.0)      # mixed layouts


def test_shapes_outside_kernel_limits_take_the_eager_chain():
    """65x65 latents (HW % 8 != 0), 12 channels (C % 8 != 0): the wrappers run the reference's op chain instead."""
    from sta import fused
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 32, 5, 5, generator=g).bfloat16().cuda()
    w = torch.ones(32).bfloat16().cuda()
    y = fused.groupnorm_silu(x, w, w * 0, 4, 1e-5)
    _close(y, F.silu(F.group_norm(x.float().cpu(), 4)), torch.bfloat16)
    xc = torch.randn(2, 12, 8, 8, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    w12 = torch.ones(12).bfloat16().cuda()
    _close(fused.groupnorm_silu(xc, w12, w12 * 0, 3, 1e-5, silu=False), F.group_norm(xc.float().cpu(), 3), torch.bfloat16)
    _close(fused.add_bias_nchw(x, x, w), 2 * x.float().cpu() + 1.0, torch.bfloat16, k=1.0)
    h = torch.randn(6, 24, generator=g).bfloat16().cuda()
    a, gate = h.float().cpu().chunk(2, dim=-1)
    _close(fused.geglu(h), a * F.gelu(gate), torch.bfloat16)
    t = torch.randn(6, 12, generator=g).bfloat16().cuda()
    s, y = fused.add_layernorm(t, t, None, w12, w12 * 0, 1e-5)
    _close(s, 2 * t.float().cpu(), torch.bfloat16, k=1.0)
    _close(y, F.layer_norm(2 * t.float().cpu(), (12,)), torch.bfloat16)


def test_fused_error_convention():
    from sta import fused
    from sta import lib
    x = torch.randn(2, 30, 5, 5, device="cuda", dtype=torch.bfloat16)          # HW = 25: not a multiple of 8
    w = torch.ones(30, device="cuda", dtype=torch.bfloat16)
    rc = lib.load().sta_groupnorm_silu(x.data_ptr(), 0, w.data_ptr(), w.data_ptr(), x.data_ptr(), 2, 30, 25, 3, 1e-5, 1, 0, 0)
    assert rc == -1 and b"groupnorm" in lib.load().sta_last_error()            # the C-ABI rejects it; the wrapper never calls it
    assert not fused.usable(torch.randn(4, 8))                                  # CPU tensors never take the fused path


# ---- input gradients for the tracked epochs (csrc/sta_unet_bwd.hip) vs fp32 autograd of the reference's op chain --------
def _grad_close(got, ref, dtype, k=4.0):
    """Gradient of a 16-bit chain: the error scales with the gradient's magnitude over the tensor (sums of many products)."""
    err = (got.float().cpu() - ref).abs()
    tol = k * EPS[dtype] * (ref.abs() + ref.abs().mean() + 1e-6)
    assert (err <= tol).all(), "max err %.4g at tol %.4g (ref max %.4g)" % (err.max().item(), tol[err.argmax()].item() if err.numel() else 0, ref.abs().max().item())


@pytest.mark.parametrize("B,C,H,G", [(2, 320, 64, 32), (3, 640, 32, 32), (2, 1280, 16, 32), (2, 2560, 8, 32), (2, 64, 12, 8), (2, 960, 64, 32), (2, 128, 64, 32)])   # last: VAE decoder, 4 channels per group
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_add,silu", [(False, True), (True, True), (False, False)])
def test_groupnorm_silu_tracked_gradient(B, C, H, G, dtype, with_add, silu):
    from sta import fused
    g = torch.Generator().manual_seed(B * C + H + 1)
    x = (torch.randn(B, C, H, H, generator=g) * 1.7 + 0.3).to(dtype)
    w = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dtype)
    b = (0.2 * torch.randn(C, generator=g)).to(dtype)
    add = torch.randn(B, C, generator=g) if with_add else None
    dy = torch.randn(B, C, H, H, generator=g).to(dtype)
    xr = x.float().requires_grad_(True)
    ref = F.group_norm(xr + (add[:, :, None, None] if with_add else 0.0), G, w.float(), b.float(), 1e-5)
    ref = F.silu(ref) if silu else ref
    ref.backward(dy.float())
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with fused.tracked():
        assert fused.tracked_usable(xg)
        y = fused.groupnorm_silu_tracked(xg, w.cuda(), b.cuda(), G, 1e-5, add=None if add is None else add.cuda(), silu=silu)
    assert fused.is_nhwc(y) and y.grad_fn is not None and "GroupNormSiLUFn" in type(y.grad_fn).__name__
    y.backward(dy.cuda().contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    _close(y.detach(), ref.detach(), dtype)
    _grad_close(xg.grad, xr.grad, dtype)


@pytest.mark.parametrize("R,D", [(4096, 1280), (1000, 2560), (64, 5120), (7, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_geglu_tracked_gradient(R, D, dtype):
    from sta import fused
    g = torch.Generator().manual_seed(R + D + 1)
    h = (torch.randn(R, 2 * D, generator=g) * 2).to(dtype)
    dy = torch.randn(R, D, generator=g).to(dtype)
    hr = h.float().requires_grad_(True)
    a, gate = hr.chunk(2, dim=-1)
    (a * F.gelu(gate)).backward(dy.float())
    hg = h.cuda().requires_grad_(True)
    with fused.tracked():
        y = fused.geglu_tracked(hg)
    assert "GegluFn" in type(y.grad_fn).__name__
    y.backward(dy.cuda())
    torch.cuda.synchronize()
    _close(hg.grad, hr.grad, dtype)


@pytest.mark.parametrize("R,C", [(8192, 320), (2048, 640), (513, 1280), (3, 2048), (5, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_f", [True, False])
def test_add_layernorm_tracked_gradient(R, C, dtype, with_f):
    """(s, y) = (x + f + bias, LN(s)) with BOTH outputs used downstream, as in the transformer block."""
    from sta import fused
    g = torch.Generator().manual_seed(R + C + 1)
    x = torch.randn(R, C, generator=g).to(dtype)
    f = torch.randn(R, C, generator=g).to(dtype) if with_f else None
    bias = (0.1 * torch.randn(C, generator=g)).to(dtype)
    w = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dtype)
    b = (0.2 * torch.randn(C, generator=g)).to(dtype)
    ds, dy = torch.randn(R, C, generator=g).to(dtype), torch.randn(R, C, generator=g).to(dtype)
    xr = x.float().requires_grad_(True)
    fr = f.float().requires_grad_(True) if with_f else None
    sr = ((xr + fr) if with_f else xr) + bias.float()
    yr = F.layer_norm(sr, (C,), w.float(), b.float(), 1e-5)
    torch.autograd.backward([sr, yr], [ds.float(), dy.float()])
    xg = x.cuda().requires_grad_(True)
    fg = f.cuda().requires_grad_(True) if with_f else None
    with fused.tracked():
        s, y = fused.add_layernorm_tracked(xg, fg, bias.cuda(), w.cuda(), b.cuda(), 1e-5)
    torch.autograd.backward([s, y], [ds.cuda(), dy.cuda()])
    torch.cuda.synchronize()
    _close(y.detach(), yr.detach(), dtype, k=4.0)
    _grad_close(xg.grad, xr.grad, dtype)
    if with_f:
        assert torch.equal(fg.grad, xg.grad)


def test_add_bias_tracked_gradient():
    from sta import fused
    g = torch.Generator().manual_seed(5)
    a = torch.randn(2, 64, 8, 8, generator=g).half().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(2, 64, 8, 8, generator=g).half().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bias = torch.randn(64, generator=g).half().cuda()
    dy = torch.randn(2, 64, 8, 8, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    with fused.tracked():
        y = fused.add_bias_tracked(a, b, bias)
    y.backward(dy)
    _close(y.detach(), (a.float() + b.float() + bias.float()[None, :, None, None]).detach().cpu(), torch.float16)
    assert torch.equal(a.grad, dy) and torch.equal(b.grad, dy)


@pytest.mark.parametrize("B,Cin,Cout,H,W,up2", [
    (8, 64, 160, 8, 32, False),      # one 8 x 32 tile per image, one part, two channel steps
    (2, 320, 320, 32, 32, False),    # four tiles per image, two parts (level 1 geometry)
    (1, 128, 160, 64, 64, False),    # 16 tiles of one image: every border case of the 8 x 32 tiling
    (8, 192, 320, 16, 16, False),    # the 16 x 16 tile (level 2 geometry), six channel steps
    (2, 64, 160, 32, 32, True),      # Upsample.conv: 16 x 16 input read through (y >> 1, x >> 1)
    (8, 128, 160, 16, 16, True),     # 8 x 8 input upsampled into the 16 x 16 tile
    (1, 64, 320, 16, 16, False),     # one tile, two parts: the plain item order (tile count not a multiple of 8)
    (3, 64, 160, 32, 32, True),      # 12 tiles, plain order, upsampled input
    (2, 128, 128, 64, 64, False),    # the VAE decoder's channel counts: a workgroup owns 128 output channels (4 row tiles per wave)
    (2, 64, 256, 16, 32, True),      # ... its Upsample convolution, two parts
    (4, 64, 512, 16, 16, False),     # ... on the 16 x 16 tile, four parts
    (4, 64, 160, 8, 8, False),       # the 8 x 8 level: two whole images per tile, two pixel segments per wave
    (3, 128, 320, 8, 8, False),      # ... odd batch: the last tile holds one image
    (8, 64, 128, 8, 8, False),       # ... 128-channel parts
    (16, 640, 320, 64, 64, False),   # more items than workgroups: the persistent loop crosses tiles (512 items)
])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("w_nhwc,epi", [(True, False), (False, True)])
def test_conv3x3_nhwc(B, Cin, Cout, H, W, up2, dtype, w_nhwc, epi):
    """csrc/sta_conv.hip against an fp32 convolution of the same 16-bit operands (ResBlock / Upsample convolutions,
    openaimodel.py:163-275, :107-120). fp32 accumulation over 9 * Cin products: the error is the 16-bit rounding of the result."""
    from sta import fused
    g = torch.Generator().manual_seed(Cin + Cout + H)
    Hs, Ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Cin, Hs, Ws, generator=g).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dtype)
    xi = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up2 else x.float()
    bias = (0.5 * torch.randn(Cout, generator=g)).to(dtype) if epi else None
    res = torch.randn(B, Cout, H, W, generator=g).to(dtype) if epi else None
    ref = F.conv2d(xi, w.float(), None if bias is None else bias.float(), 1, 1) + (res.float() if epi else 0.0)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last) if w_nhwc else w.cuda()
    with torch.no_grad():
        assert fused.conv3x3_supported(xd, wd, up2=up2)
        got = fused.conv3x3_nhwc(xd, fused.pack_conv3x3_weight(wd), Cout, up2=up2, bias=None if bias is None else bias.cuda(),
                                 res=None if res is None else res.cuda().contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    _close(got, ref, dtype, k=2.0)


def test_conv3x3_batch_split_below_the_descriptor_limit(monkeypatch):
    """A launch addresses its output through one 32-bit buffer descriptor; larger batches (the decoder's 256-channel 512 x 512
    upsample convolution at 32 images: 4.3 GB) go in several launches over slices of the same tensors."""
    from sta import fused
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 64, 16, 32, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(160, 64, 3, 3, generator=g) / 24.0).half().cuda()
    res = torch.randn(5, 160, 16, 32, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        wp = fused.pack_conv3x3_weight(w)
        one = fused.conv3x3_nhwc(x, wp, 160, res=res)
        monkeypatch.setattr(fused, "CONV_MAX_BYTES", 2 * 16 * 32 * 160 * 2)       # two images per launch: 2 + 2 + 1
        split = fused.conv3x3_nhwc(x, wp, 160, res=res)
    assert torch.equal(one, split)


def test_conv3x3_unsupported_geometries_are_refused():
    from sta import fused, lib
    L = lib.load()
    assert not L.sta_conv3x3_nhwc_supported(64, 4, 4, 1280, 1280) and not L.sta_conv3x3_nhwc_supported(64, 12, 12, 320, 320)
    assert not L.sta_conv3x3_nhwc_supported(64, 64, 64, 4, 320)         # conv_in
    assert not L.sta_conv3x3_nhwc_supported(64, 64, 64, 320, 4)         # conv_out
    assert L.sta_conv3x3_nhwc_supported(64, 64, 64, 960, 320) and L.sta_conv3x3_nhwc_supported(64, 16, 16, 2560, 1280)
    assert L.sta_conv3x3_nhwc_supported(64, 8, 8, 2560, 1280)
    x = torch.zeros(64, 320, 12, 12, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(320, 320, 3, 3, device="cuda", dtype=torch.float16)
    assert not fused.conv3x3_supported(x, w)
    with torch.no_grad():      # few work items: the library is faster there (tools/conv_bench.py --batch 2)
        x2 = torch.zeros(2, 1280, 16, 16, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        w2 = torch.zeros(1280, 1280, 3, 3, device="cuda", dtype=torch.float16)
        assert fused.conv3x3_supported(x2, w2)
        fused.CONV_MIN_ITEMS = 64
        assert not fused.conv3x3_supported(x2, w2) and fused.conv3x3_work_items(2, 16, 16, 1280) == 16
        assert fused.conv3x3_supported(torch.zeros(64, 1280, 16, 16, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last), w2)
    z = torch.zeros(8192, dtype=torch.uint8, device="cuda")
    assert L.sta_conv3x3_nhwc(x.data_ptr(), w.data_ptr(), z.data_ptr(), None, None, x.data_ptr(), None, 64, 12, 12, 320, 320, 0, 1, None) != 0
    assert "unsupported geometry" in lib.last_error()


@pytest.mark.parametrize("R,K,N", [
    (4096, 320, 320),        # 16 row tiles x 2 parts, five steps (odd: the ring slot parity alternates between tiles)
    (8192, 640, 640),        # level-1 projection shape
    (4096 + 48, 64, 160),    # one step; rows past R in the last tile
    (20000, 960, 320),       # 15 steps, ragged rows, more items than a small grid in plain order
    (70000, 320, 1280),      # 274 row tiles x 8 parts: the persistent loop crosses tiles with an odd step count
    (4096, 1280, 256),       # 128-column parts
])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("epi", [False, True])
def test_linear_rows(R, K, N, dtype, epi):
    """csrc/sta_gemm.hip against an fp32 product of the same 16-bit operands (Linear layers / 1x1 convolutions of the transformer
    blocks: attention.py:158-215, :322-333): fp32 accumulation over K products, error = the 16-bit rounding of the result."""
    from sta import fused
    g = torch.Generator().manual_seed(R + K + N)
    x = torch.randn(R, K, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
    bias = (0.5 * torch.randn(N, generator=g)).to(dtype) if epi else None
    res = torch.randn(R, N, generator=g).to(dtype) if epi else None
    ref = F.linear(x.float(), w.float(), None if bias is None else bias.float()) + (res.float() if epi else 0.0)
    with torch.no_grad():
        xd, wd = x.cuda(), w.cuda()
        assert fused.linear_rows_supported(xd, wd)
        got = fused.linear_rows(xd, fused.pack_linear_weight(wd), N, bias=None if bias is None else bias.cuda(), res=None if res is None else res.cuda())
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    _close(got, ref, dtype, k=2.0)


def test_linear_rows_conv1x1_weight_and_refusals():
    from sta import fused, lib
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 1024, 320, generator=g).half().cuda()
    wc = (torch.randn(320, 320, 1, 1, generator=g) / 18.0).half().cuda().contiguous(memory_format=torch.channels_last)   # a 1x1 Conv2d weight
    with torch.no_grad():
        got = fused.linear_rows(x, fused.pack_linear_weight(wc[:, :, 0, 0]), 320)
        assert not fused.linear_rows_supported(x[:, :8], wc[:, :, 0, 0])          # below LINEAR_MIN_ROWS
        assert not fused.linear_rows_supported(x, torch.zeros(100, 320, device="cuda", dtype=torch.float16))
    _close(got, F.linear(x.float().cpu(), wc[:, :, 0, 0].float().cpu()), torch.float16, k=2.0)
    assert not lib.load().sta_linear_rows_supported(4096, 100, 320) and not lib.load().sta_linear_rows_supported(1 << 24, 320, 320)


@pytest.mark.parametrize("B,Ca,Cb,H,G", [(2, 640, 320, 32, 32), (2, 320, 320, 64, 32), (3, 1280, 640, 16, 32), (2, 64, 64, 8, 32), (2, 1280, 1280, 8, 32)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_groupnorm_silu_of_a_concatenation_read_in_place(B, Ca, Cb, H, G, dtype):
    """GroupNorm + SiLU of cat([xa, xb], dim=1) (the UNet's output blocks, reference openaimodel.py:740 + ResBlock.in_layers) with both
    tensors read in place, against the same kernel on the materialised concatenation (groups may straddle the seam: 960 / 32) to one
    ulp (the statistics go through LDS atomics: the summation order is not reproducible) and against the fp32 chain."""
    from sta import fused
    g = torch.Generator().manual_seed(Ca + Cb + H)
    xa = (torch.randn(B, Ca, H, H, generator=g) * 1.3 + 0.2).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    xb = (torch.randn(B, Cb, H, H, generator=g) * 0.7 - 0.4).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    w = (1.0 + 0.2 * torch.randn(Ca + Cb, generator=g)).to(dtype).cuda()
    b = (0.2 * torch.randn(Ca + Cb, generator=g)).to(dtype).cuda()
    with torch.no_grad():
        got = fused.groupnorm_silu_cat(xa, xb, w, b, G, 1e-5)
        ref = fused.groupnorm_silu(torch.cat([xa, xb], dim=1), w, b, G, 1e-5)
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    _close(got, ref.float().cpu(), dtype, k=2.0)
    ref32 = F.silu(F.group_norm(torch.cat([xa, xb], dim=1).float().cpu(), G, w.float().cpu(), b.float().cpu(), 1e-5))
    _close(got, ref32, dtype, k=3.0)


@pytest.mark.parametrize("R,Ka,Kb,N", [(4096, 320, 320, 320), (8192, 640, 320, 320), (5000, 1280, 640, 640), (4096, 64, 64, 160), (4096, 1280, 1280, 1280)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_rows_of_a_concatenation_read_in_place(R, Ka, Kb, N, dtype):
    """The 1x1 skip convolution over cat([h, skip]) as one row GEMM over both tensors: bit-identical to the same kernel on the
    materialised concatenation (same products in the same order)."""
    from sta import fused
    g = torch.Generator().manual_seed(R + Ka + N)
    xa = torch.randn(R, Ka, generator=g).to(dtype).cuda()
    xb = torch.randn(R, Kb, generator=g).to(dtype).cuda()
    w = (torch.randn(N, Ka + Kb, generator=g) / (Ka + Kb) ** 0.5).to(dtype).cuda()
    bias = (0.3 * torch.randn(N, generator=g)).to(dtype).cuda()
    with torch.no_grad():
        wp = fused.pack_linear_weight(w)
        got = fused.linear_rows_cat(xa, xb, wp, N, bias=bias)
        ref = fused.linear_rows(torch.cat([xa, xb], dim=1), wp, N, bias=bias)
    assert torch.equal(got, ref)
    _close(got, F.linear(torch.cat([xa, xb], dim=1).float().cpu(), w.float().cpu(), bias.float().cpu()), dtype, k=2.0)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 320, 320, 32, 32), (2, 640, 320, 16, 16), (1, 128, 320, 64, 64), (4, 320, 640, 8, 8), (2, 960, 320, 16, 32)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv3x3_tracked_gradient(B, Cin, Cout, H, W, dtype):
    """The differentiable form for the tracked epochs (frozen weights): forward and INPUT GRADIENT both on csrc/sta_conv.hip (the
    gradient is the convolution with the channel axes exchanged and the taps mirrored) against fp32 autograd of the same operands."""
    from sta import fused
    g = torch.Generator().manual_seed(Cin + Cout + H + 7)
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dtype)
    dy = torch.randn(B, Cout, H, W, generator=g).to(dtype)
    xr = x.float().requires_grad_(True)
    ref = F.conv2d(xr, w.float(), None, 1, 1)
    ref.backward(dy.float())
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1).to("cuda", dtype)
    with torch.no_grad():
        conv.weight.copy_(w)
    for p_ in conv.parameters():
        p_.requires_grad_(False)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with fused.tracked():
        assert fused.conv3x3_tracked_supported(xg, conv.weight)
        y = fused.conv3x3_tracked(conv, conv, xg)
    assert y.grad_fn is not None and "Conv3x3Fn" in type(y.grad_fn).__name__
    y.backward(dy.cuda())                                     # an NCHW gradient: made NHWC inside
    torch.cuda.synchronize()
    _close(y.detach(), ref.detach(), dtype, k=2.0)
    _close(xg.grad, xr.grad, dtype, k=2.0)


@pytest.mark.parametrize("B,Cin,Cout,H,W,up2", [(2, 64, 320, 32, 32, False), (3, 128, 160, 16, 16, False), (5, 64, 160, 8, 8, False), (2, 64, 256, 16, 32, True),
                                               (16, 320, 320, 64, 64, False)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv3x3_epilogue_statistics_feed_the_groupnorm(B, Cin, Cout, H, W, up2, dtype):
    """The convolution's epilogue accumulates every output channel's sum / sum of squares of the values it stores (fp32 atomics); the
    GroupNorm that consumes the tensor takes them instead of running its statistics pass (ResBlock: conv -> GroupNorm, openaimodel.py
    in_layers / out_layers). Statistics against torch sums of the stored tensor; the one-pass GroupNorm (with and without the
    per-(b, c) pre-add) against the two-pass kernel on the same tensor, to two ulps."""
    from sta import fused
    g = torch.Generator().manual_seed(B + Cin + Cout + H)
    Hs, Ws = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, Cin, Hs, Ws, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dtype).cuda()
    bias = (0.5 * torch.randn(Cout, generator=g)).to(dtype).cuda()
    res = torch.randn(B, Cout, H, W, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    gw = (1.0 + 0.2 * torch.randn(Cout, generator=g)).to(dtype).cuda()
    gb = (0.2 * torch.randn(Cout, generator=g)).to(dtype).cuda()
    add = torch.randn(B, Cout, generator=g).cuda()
    with torch.no_grad():
        wp = fused.pack_conv3x3_weight(w)
        y = fused.conv3x3_nhwc(x, wp, Cout, up2=up2, bias=bias, res=res, stats=True)
        plain = fused.conv3x3_nhwc(x, wp, Cout, up2=up2, bias=bias, res=res)
        assert torch.equal(y, plain) and not hasattr(plain, "_sta_stats")
        st = fused._producer_stats(y)
        yf = y.float()
        ref = torch.stack([yf.sum(dim=(2, 3)), (yf * yf).sum(dim=(2, 3))], dim=-1)
        assert st.shape == (B, Cout, 2)
        assert ((st - ref).abs() <= 2e-3 * ref.abs() + 2e-2).all(), (st - ref).abs().max().item()
        for a_ in (None, add):
            one = fused.groupnorm_silu(y, gw, gb, 32, 1e-5, add=a_)
            two = fused.groupnorm_silu(plain, gw, gb, 32, 1e-5, add=a_)
            _close(one, two.float().cpu(), dtype, k=2.0)


def test_linear_rows_epilogue_statistics():
    from sta import fused
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 1024, 320, generator=g).half().cuda()
    w = (torch.randn(320, 320, generator=g) / 18.0).half().cuda()
    with torch.no_grad():
        wp = fused.pack_linear_weight(w)
        y = fused.linear_rows(x, wp, 320, stats_rows=1024)
        assert torch.equal(y, fused.linear_rows(x, wp, 320))
        yf = y.float()
        ref = torch.stack([yf.sum(dim=1), (yf * yf).sum(dim=1)], dim=-1)
        assert ((fused._producer_stats(y) - ref).abs() <= 2e-3 * ref.abs() + 2e-2).all()
        y.add_(1.0)                                     # an in-place update: the producer's sums no longer describe y and are ignored
        assert fused._producer_stats(y) is None
        assert not hasattr(fused.linear_rows(x, wp, 320, stats_rows=100), "_sta_stats")      # not a multiple of 256: no statistics
