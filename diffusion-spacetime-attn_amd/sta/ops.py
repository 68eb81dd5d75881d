"""Host side of the fused spatial-temporal cross-attention: tensors in, C-ABI calls out.

Mirrors (reference paths under attention_optimization/stable-diffusion/):
  * disc masks                      ldm/modules/attention.py:251-262
  * K/V re-layout per context       ldm/modules/attention.py:180-183   (-> sta_xattn_pack_kv, once per prompt)
  * (K+1) x attn2 + masked blend    ldm/modules/attention.py:175-197, 278-294   (-> sta_xattn_fwd)
  * its gradient w.r.t. x and coef  ldm/modules/diffusionmodules/util.py:132-145 (-> sta_xattn_bwd)

PyTorch is used for device memory, the current HIP stream and autograd plumbing only; all arithmetic
of the path happens in csrc/sta_xattn.hip. Nothing here falls back to eager PyTorch.
"""
import os
import torch

from . import lib as _lib

_DTYPES = {torch.bfloat16: _lib.STA_BF16, torch.float16: _lib.STA_F16}


def _dtype_code(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError("fused cross-attention supports bf16/fp16 tensors, got %s" % t.dtype) from None


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def disc_masks(centres, dim, radius_sq=0.04):
    """[K, dim*dim] uint8 CPU tensor, 1 where pixel (row r, col c) lies inside object i's disc.

    Bit-exact float32 restatement of attention.py:251-262: ``x`` runs along columns, ``y`` along
    rows, coordinates are ``arange(dim)/dim`` and the test is ``dx^2 + dy^2 < 0.04`` (r = 0.2).
    """
    axis = torch.arange(dim, dtype=torch.float32) / dim
    out = torch.zeros((len(centres), dim * dim), dtype=torch.uint8)
    for i, (cx, cy) in enumerate(centres):
        dx2 = (axis - cx) ** 2
        dy2 = (axis - cy) ** 2
        inside = (dx2[None, :] + dy2[:, None]) < radius_sq
        out[i] = inside.reshape(-1).to(torch.uint8)
    return out


def mask_bits(mask_kn):
    """[K, N] 0/1 masks -> [N] uint8 bit field (bit i = inside disc i), the layout the kernels read."""
    K = mask_kn.shape[0]
    if K > _lib.MAX_OBJECTS:
        raise ValueError("at most %d objects are supported, got %d" % (_lib.MAX_OBJECTS, K))
    weights = (2 ** torch.arange(K, dtype=torch.int32, device=mask_kn.device)).view(K, 1)
    return (mask_kn.to(torch.int32).ne(0).to(torch.int32) * weights).sum(0).to(torch.uint8)


def disc_mask_bits(centres, dim):
    """Disc masks of all objects as the [N] uint8 bit field (CPU tensor)."""
    return mask_bits(disc_masks(centres, dim))


class PackedKV:
    """Fragment image of the projected keys/values of all contexts of one transformer block, for
    `n_img` images (prompts) that share the object count: n_ctx = K + 2 contexts per image."""

    __slots__ = ("buf", "n_ctx", "heads", "M", "C", "dtype", "n_img")

    def __init__(self, buf, n_ctx, heads, M, C, dtype, n_img=1):
        self.buf, self.n_ctx, self.heads, self.M, self.C, self.dtype, self.n_img = buf, n_ctx, heads, M, C, dtype, n_img


def pack_kv(k, v, heads, out=None, n_img=1):
    """k, v: [n_img * n_ctx, M, C], image-major; per image ctx 0 = "", 1 = global prompt, 2+i = local
    prompt i -> PackedKV. `out`: a PackedKV of the same shape to refill in place (stable device address)."""
    if k.shape != v.shape or k.dim() != 3:
        raise ValueError("k and v must both be [n_img * n_ctx, M, C], got %s and %s" % (tuple(k.shape), tuple(v.shape)))
    if not k.is_cuda:
        raise RuntimeError("pack_kv needs CUDA/HIP tensors (there is no CPU path)")
    total, M, C = k.shape
    if C % heads:
        raise ValueError("C=%d is not divisible by heads=%d" % (C, heads))
    if n_img < 1 or total % n_img:
        raise ValueError("%d contexts do not split over n_img=%d images" % (total, n_img))
    n_ctx = total // n_img
    L = _lib.load()
    k, v = k.contiguous(), v.contiguous()
    nbytes = L.sta_xattn_packed_kv_bytes(total, heads, C // heads)
    if nbytes == 0:
        raise ValueError("unsupported head dim %d (need d %% 8 == 0 and d <= %d)" % (C // heads, _lib.MAX_HEAD_DIM))
    if out is not None and (out.n_ctx, out.heads, out.M, out.C, out.dtype, out.n_img) == (n_ctx, heads, M, C, k.dtype, n_img) \
            and out.buf.device == k.device:
        buf = out.buf
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=k.device)
    _lib.check(L.sta_xattn_pack_kv(k.data_ptr(), v.data_ptr(), buf.data_ptr(), total, M, C, heads,
                                   _dtype_code(k), _stream(k)), "sta_xattn_pack_kv")
    return PackedKV(buf, n_ctx, heads, M, C, k.dtype, n_img)


def _check_inputs(q, packed, mask, coef):
    I = packed.n_img
    if q.dim() != 3 or q.shape[0] != 2 * I:
        raise ValueError("q must be [2 * n_img, N, C] (per image: uncond row, cond row) with n_img=%d, got %s"
                         % (I, tuple(q.shape)))
    if not q.is_cuda:
        raise RuntimeError("fused cross-attention needs CUDA/HIP tensors (there is no CPU path)")
    if q.dtype != packed.dtype:
        raise TypeError("q is %s but K/V were packed as %s" % (q.dtype, packed.dtype))
    N, C = q.shape[1], q.shape[2]
    if C != packed.C:
        raise ValueError("q has C=%d but K/V were packed with C=%d" % (C, packed.C))
    K = packed.n_ctx - 2
    if K < 0:
        raise ValueError("need at least the unconditional and the global context")
    if K > 0:
        if mask is None or coef is None:
            raise ValueError("mask and coef are required when local contexts are present")
        if mask.numel() != I * N or mask.dtype != torch.uint8:
            raise ValueError("mask must be the uint8 bit field [n_img=%d, N=%d] (see mask_bits), got %s %s"
                             % (I, N, mask.dtype, tuple(mask.shape)))
        if coef.numel() != I * K:
            raise ValueError("coef must have n_img*K=%d elements, got %d" % (I * K, coef.numel()))
    return I, N, C, K


SELFATTN_ENABLED = True      # A/B switch for tools/ (False: attn1 through PyTorch SDPA); nothing reads the environment

# bench.py's roofline legs: when set to a list, every forward launch (and every sta_xattn_bwd call: kind "bwd") is bracketed IN SITU (inside a real UNet call) by its
# own HIP-event pair on the launch stream and (kind, n_img, N, C, K, e0, e1, relaunch) is appended; `relaunch()` re-issues
# exactly that C-ABI call on the same tensors (for warm, back-to-back timing beside the in-situ figure).
LAUNCH_LOG = None


def _logged(kind, I, N, C, K, launch):
    if LAUNCH_LOG is None:
        launch()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    LAUNCH_LOG.append((kind, I, N, C, K, e0, e1, launch))


def xattn_forward(q, packed, mask, coef, scale, want_maps=False):
    """Raw forward call (no autograd). Returns (out [2I,N,C], maps [I,K+2,heads,N,M] fp32 or None)."""
    I, N, C, K = _check_inputs(q, packed, mask, coef)
    L = _lib.load()
    q = q.contiguous()
    coef32 = coef.detach().to(torch.float32).contiguous() if K else None
    maskc = mask.contiguous() if K else None
    out = torch.empty_like(q)
    maps = torch.empty((I, K + 2, packed.heads, N, packed.M), dtype=torch.float32, device=q.device) if want_maps else None
    def launch():
        _lib.check(L.sta_xattn_fwd(q.data_ptr(), packed.buf.data_ptr(), _ptr(maskc), _ptr(coef32), out.data_ptr(),
                                   _ptr(maps), I, N, C, packed.heads, packed.M, K, float(scale), _dtype_code(q),
                                   _stream(q)), "sta_xattn_fwd")
    _logged("attn", I, N, C, K, launch)
    if maps is not None and I == 1:
        maps = maps[0]
    return out, maps


def xattn_backward(q, packed, mask, coef, dout, scale):
    """Raw backward call. Returns (dq [2I,N,C], dcoef [I*K] fp32)."""
    I, N, C, K = _check_inputs(q, packed, mask, coef)
    L = _lib.load()
    q, dout = q.contiguous(), dout.to(q.dtype).contiguous()
    coef32 = coef.detach().to(torch.float32).contiguous() if K else None
    maskc = mask.contiguous() if K else None
    dq = torch.empty_like(q)
    dcoef = torch.empty(I * K, dtype=torch.float32, device=q.device)
    ws = torch.empty(L.sta_xattn_bwd_workspace_bytes(I, N, packed.heads, K), dtype=torch.uint8, device=q.device)
    def launch():
        _lib.check(L.sta_xattn_bwd(q.data_ptr(), packed.buf.data_ptr(), _ptr(maskc), _ptr(coef32), dout.data_ptr(),
                                   dq.data_ptr(), _ptr(dcoef) if K else 0, ws.data_ptr(), I, N, C, packed.heads,
                                   packed.M, K, float(scale), _dtype_code(q), _stream(q)), "sta_xattn_bwd")
    _logged("bwd", I, N, C, K, launch)
    return dq, dcoef


# ---------------------------------------------------------------------------------------------------
# forward with the query projection inside (sta_xattn_fwd_proj; inference only)
# ---------------------------------------------------------------------------------------------------
PROJ_MIN_WORKGROUPS = 256     # 128-pixel workgroup tiles x heads x images from which the projection-fused kernel is used


PROJ_LL2_IN_MODEL = True      # does the model take the locals-from-L2 variant (SD-v1 level 1) where the launch is large enough?
                              # (measured in situ against the to_q GEMM + sta_xattn_fwd: profiles/r05_level1_proj.md)


PROJ_WQS_IN_MODEL = False     # does the model take the STREAMED-Wq variant (SD-v1 levels 2 / mid, d = 160)? Built and oracle-exact in round 6, and measured
                              # SLOWER than the library to_q GEMM + sta_xattn_fwd it would replace (64 images, N = 256: 241 vs 194 us warm, 292 vs 219 us from
                              # HBM; N = 64: 91 vs 66): one Wq fragment read feeds two MFMAs and every chunk ends in a barrier (profiles/r06_level2_proj.md)


def proj_streams_wq(C, heads):
    """Shapes whose Wq slice does not fit a CU: sta_xattn_fwd_proj streams it through an LDS ring (csrc/sta_xattn_proj.hip)."""
    return C % heads == 0 and 144 < C // heads <= 160 and C % 128 == 0


def proj_locals_from_l2(C, heads, M, K):
    """Shapes sta_xattn_fwd_proj takes with only Wq + the two mandatory contexts resident (local contexts read from L2)."""
    return bool(_lib.load().sta_xattn_fwd_proj_locals_from_l2(C, heads, M, K))


def proj_supported(C, heads, M, K, N=None, n_img=1):
    """Shapes sta_xattn_fwd_proj takes (Wq slice + all contexts resident in LDS, or — SD-v1 level 1 — the local contexts left
    in L2) and, when N is given (the model's question), launches large enough that its one-workgroup-per-CU structure fills the chip."""
    if not _lib.load().sta_xattn_fwd_proj_supported(C, heads, M, K):
        return False
    if N is None:
        return True
    if proj_locals_from_l2(C, heads, M, K) and not PROJ_LL2_IN_MODEL:
        return False
    if proj_streams_wq(C, heads) and not PROJ_WQS_IN_MODEL:
        return False
    return ((N + 127) // 128) * heads * n_img >= PROJ_MIN_WORKGROUPS


def pack_wq(weight, heads):
    """to_q.weight [C, C] -> per-head MFMA fragment image (uint8 tensor); once per model."""
    C = weight.shape[0]
    if weight.shape != (C, C) or not weight.is_cuda:
        raise ValueError("to_q.weight must be a square CUDA matrix, got %s" % (tuple(weight.shape),))
    L = _lib.load()
    nbytes = L.sta_xattn_packed_wq_bytes(C, heads)
    if nbytes == 0:
        raise ValueError("unsupported projection shape C=%d heads=%d" % (C, heads))
    w = weight.detach().contiguous()
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(L.sta_xattn_pack_wq(w.data_ptr(), buf.data_ptr(), C, heads, _dtype_code(w), _stream(w)), "sta_xattn_pack_wq")
    return buf


def pack_kv_proj(k, v, heads, out=None, n_img=1):
    """Like pack_kv, for sta_xattn_fwd_proj: forward-only image, K fragments in projected-query order."""
    if k.shape != v.shape or k.dim() != 3 or not k.is_cuda:
        raise ValueError("k and v must both be CUDA tensors [n_img * n_ctx, M, C]")
    total, M, C = k.shape
    if C % heads or n_img < 1 or total % n_img:
        raise ValueError("bad shapes: %s, heads=%d, n_img=%d" % (tuple(k.shape), heads, n_img))
    n_ctx = total // n_img
    L = _lib.load()
    k, v = k.contiguous(), v.contiguous()
    nbytes = L.sta_xattn_packed_kv_proj_bytes(total, heads, C // heads)
    if nbytes == 0:
        raise ValueError("unsupported head dim %d" % (C // heads))
    if out is not None and (out.n_ctx, out.heads, out.M, out.C, out.dtype, out.n_img) == (n_ctx, heads, M, C, k.dtype, n_img) \
            and out.buf.device == k.device and out.buf.numel() == nbytes:
        buf = out.buf
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=k.device)
    _lib.check(L.sta_xattn_pack_kv_proj(k.data_ptr(), v.data_ptr(), buf.data_ptr(), total, M, C, heads,
                                        _dtype_code(k), _stream(k)), "sta_xattn_pack_kv_proj")
    return PackedKV(buf, n_ctx, heads, M, C, k.dtype, n_img)


def _ensure_toolchain_checked():
    """Run the head-pair kernel's self-check (once per process) BEFORE any producer is told to write fragment order: a library built by
    an unvalidated hipcc release whose pair kernel disagrees switches the pair kernel off (STA_OPT_PROJ_PAIR = 2), and
    sta_xattn_fwd_proj_qfrag_supported then answers no — so the block takes the row-major chain from its first call on."""
    global _TOOLCHAIN_CHECKED
    if _TOOLCHAIN_CHECKED is None:
        if _lib.toolchain_validated():
            _TOOLCHAIN_CHECKED = True
        elif torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            toolchain_self_check()


def proj_qfrag_supported(C, heads, M, K, N, n_img):
    """Launches whose y the head-pair kernel reads in query-fragment order (sta_xattn_fwd_proj_qfrag_supported). This is where a block
    decides its layouts (ldm.modules.attention prepare step), so the toolchain self-check runs here, not at the first launch."""
    _ensure_toolchain_checked()
    return bool(_lib.load().sta_xattn_fwd_proj_qfrag_supported(n_img, N, C, heads, M, K))


def to_qfrag(y):
    """Row-major [..., N, C] -> query-fragment order (same shape container): groups of 16 rows as C/32 fragments of 1 KiB, lane
    16 g + c of fragment s = y[16 P + c, 32 s + 8 g .. + 7]. Host-side restatement of sta_add_layernorm_qfrag's layout for tests
    and tools; the product gets the layout from the LayerNorm pass itself."""
    C = y.shape[-1]
    rows = y.numel() // C
    if C % 32 or rows % 16:
        raise ValueError("need C %% 32 == 0 and a multiple of 16 rows, got %s" % (tuple(y.shape),))
    t = y.reshape(rows // 16, 16, C // 32, 4, 8)                 # [P, c, s, g, e]
    return t.permute(0, 2, 3, 1, 4).contiguous().view(y.shape)    # [P, s, g, c, e]


def from_qfrag(yf):
    """Inverse of to_qfrag."""
    C = yf.shape[-1]
    rows = yf.numel() // C
    t = yf.reshape(rows // 16, C // 32, 4, 16, 8)                # [P, s, g, c, e]
    return t.permute(0, 3, 1, 2, 4).contiguous().view(yf.shape)


def _vrow_dim(r, hp):
    """csrc/sta_xattn_proj3.h::vrow_dim — head dim held by row r of a head's V^T image (= O^T row r of the pair kernel)."""
    return -1 if r == 40 else (r if r >= 32 else 8 * ((((r & 15) >> 2) + 3 * hp) & 3) + 4 * (r >> 4) + (r & 3))


def ofrag_channels():
    """[10, 4, 8] long tensor: channel (head * 40 + dim) behind slot j of lane row g of fragment f in OUT-FRAGMENT order
    (csrc/sta_xattn_proj3.h::ofrag_channel restated on the host; C = 320, 8 heads)."""
    t = torch.empty(10, 4, 8, dtype=torch.long)
    for f in range(10):
        for g in range(4):
            for j in range(8):
                if f < 8:
                    hp, r = f & 1, (4 * g + j if j < 4 else 16 + 4 * g + (j - 4))
                    t[f, g, j] = f * 40 + _vrow_dim(r, hp)
                else:
                    pr, hp, r = 2 * (f - 8) + (g >> 1), (0 if j < 4 else 1), 32 + 4 * (g & 1) + (j & 3)
                    t[f, g, j] = (2 * pr + hp) * 40 + _vrow_dim(r, hp)
    return t


def from_ofrag(of):
    """Out-fragment order (what sta_xattn_fwd_proj_qfrag_ofrag writes) -> row-major [..., N, 320]; tests and tools only."""
    C = of.shape[-1]
    rows = of.numel() // C
    if C != 320 or rows % 16:
        raise ValueError("out-fragment order: C = 320 and a multiple of 16 rows, got %s" % (tuple(of.shape),))
    t = of.reshape(rows // 16, 10, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(rows // 16, 16, 320)     # [P, c, (f, g, j)]
    idx = ofrag_channels().reshape(-1).to(of.device)
    out = torch.empty_like(t)
    out[..., idx] = t
    return out.reshape(of.shape)


def to_ofrag(x):
    """Inverse of from_ofrag."""
    C = x.shape[-1]
    rows = x.numel() // C
    idx = ofrag_channels().reshape(-1).to(x.device)
    t = x.reshape(rows // 16, 16, 320)[..., idx].reshape(rows // 16, 16, 10, 4, 8).permute(0, 2, 3, 1, 4)
    return t.contiguous().view(x.shape)


def proj_ofrag_supported(C, heads, dtype=torch.float16):
    """Shapes whose blended output the attention kernel can leave in out-fragment order for sta.fused.to_out_add_layernorm_ofrag:
    C = 320 with 8 heads, fp16 and bf16."""
    return bool(_lib.load().sta_to_out_ln_packed_wo_bytes(C, heads))


_TOOLCHAIN_CHECKED = None


def toolchain_self_check(force=False, device=None):
    """The head-pair kernel of level 0 (csrc/sta_xattn_proj3.hip) sits on MFMA hazard windows that hipcc does not pad by itself; the build's
    lint enforces the ones measured with hipcc 7.2. A library built by ANOTHER release (lib.toolchain_validated() is False) is therefore
    checked once per process before its first use — and always when `force`: a small level-0 problem through the pair kernel's three
    layouts (row-major, query fragments, out fragments) against the one-head-per-workgroup kernel of csrc/sta_xattn_proj.hip. On a
    mismatch the pair kernel and its fragment paths are switched off (STA_OPT_PROJ_PAIR = 2: sta_xattn_fwd_proj_qfrag_supported then
    says no and the block takes the row-major chain) with a warning, instead of producing wrong attention silently. Returns True when
    the pair kernel agrees."""
    global _TOOLCHAIN_CHECKED
    if _TOOLCHAIN_CHECKED is not None and not force:
        return _TOOLCHAIN_CHECKED
    import warnings
    _TOOLCHAIN_CHECKED = True                      # (the launches below re-enter xattn_forward_proj)
    ok = True
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    prev = _lib.get_option(_lib.OPT_PROJ_PAIR)      # a tests / tools override of the pair-kernel switch survives the check
    for dtype in (torch.float16, torch.bfloat16):
        g = torch.Generator().manual_seed(11)
        N, C, heads, K, M = 1024, 320, 8, 2, 77
        y = torch.randn(2, N, C, generator=g).to(dtype).to(dev)
        wq = (torch.randn(C, C, generator=g) / C ** 0.5).to(dtype).to(dev)
        k = (torch.randn(K + 2, M, C, generator=g) * 0.7).to(dtype).to(dev)
        v = torch.randn(K + 2, M, C, generator=g).to(dtype).to(dev)
        mb = mask_bits(torch.rand(K, N, generator=g) < 0.3).to(dev)
        coef = (torch.rand(K, generator=g) * 3 + 0.5).to(dev)
        wqf, kvp = pack_wq(wq, heads), pack_kv_proj(k, v, heads, n_img=1)
        scale = (C // heads) ** -0.5
        try:
            _lib.set_option(_lib.OPT_PROJ_PAIR, 2)
            ref = xattn_forward_proj(y, wqf, kvp, mb, coef, scale).float()
            _lib.set_option(_lib.OPT_PROJ_PAIR, 1)
            outs = [xattn_forward_proj(y, wqf, kvp, mb, coef, scale)]
            if proj_qfrag_supported(C, heads, M, K, N, 1):
                outs.append(xattn_forward_proj(to_qfrag(y), wqf, kvp, mb, coef, scale, qfrag=True))
                outs.append(from_ofrag(xattn_forward_proj(to_qfrag(y), wqf, kvp, mb, coef, scale, qfrag=True, ofrag=True)))
        finally:
            _lib.set_option(_lib.OPT_PROJ_PAIR, prev)
        eps = 2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8
        for o in outs:            # two kernels, two roundings of q (the pair kernel folds scale * log2 e into Wq): 16 eps; a broken instantiation is off by whole values
            ok = ok and bool(torch.isfinite(o).all()) and bool(((o.float() - ref).abs() <= 16 * eps * (1.0 + ref.abs())).all())
    if not ok:
        _lib.set_option(_lib.OPT_PROJ_PAIR, 2)
        warnings.warn("sta: the level-0 head-pair kernel built by '%s' disagrees with the reference kernel; it is switched off for this "
                      "process (validated toolchain: '%s...')" % (_lib.built_with(), _lib.VALIDATED_HIPCC))
    _TOOLCHAIN_CHECKED = ok
    return ok


def xattn_forward_proj(y, wq_packed, packed, mask, coef, scale, qfrag=False, ofrag=False, stats=True):
    """y [2I, N, C] = norm2(hidden) -> blended pre-projection output [2I, N, C]; the query projection happens inside
    the attention kernel (no autograd). `packed` comes from pack_kv_proj, `wq_packed` from pack_wq. `qfrag`: y is in
    query-fragment order (fused.add_layernorm(..., qfrag=True) / to_qfrag); only where proj_qfrag_supported. `ofrag`: the result
    leaves in out-fragment order for fused.to_out_add_layernorm_ofrag (from_ofrag restores row-major). `stats`: True = the per-(device,
    stream) words of proj_stats() (the head-pair kernel counts its optimistic-softmax fall-backs there and sits the optimistic path out
    when they pile up), None = no statistics and always optimistic, or an own int32 tensor of lib.P3_STATS_WORDS words."""
    I, N, C, K = _check_inputs(y, packed, mask, coef)
    L = _lib.load()
    if not qfrag:                  # (fragment-order callers asked proj_qfrag_supported first; a row-major first call checks here)
        _ensure_toolchain_checked()
    y = y.contiguous()
    coef32 = coef.detach().to(torch.float32).contiguous() if K else None
    maskc = mask.contiguous() if K else None
    out = torch.empty_like(y)
    if ofrag and not qfrag:
        raise ValueError("out-fragment order rides on the query-fragment path (qfrag=True)")
    layout = 2 if ofrag else (1 if qfrag else 0)
    stats = proj_stats(y.device) if stats is True else stats
    def launch():
        _lib.check(L.sta_xattn_fwd_proj_ex(y.data_ptr(), wq_packed.data_ptr(), packed.buf.data_ptr(), _ptr(maskc), _ptr(coef32),
                                           out.data_ptr(), I, N, C, packed.heads, packed.M, K, float(scale), _dtype_code(y), layout,
                                           _ptr(stats), _stream(y)), "sta_xattn_fwd_proj_ex")
    _logged("proj", I, N, C, K, launch)
    return out


_PROJ_STATS = {}


def proj_stats(device=None):
    """The statistics / adaptive-switch words the head-pair kernel keeps across launches (include/sta_xattn.h: sta_xattn_fwd_proj_ex), one
    buffer per device: int32 tensor of lib.P3_STATS_WORDS words — [0] launches still sitting the optimistic softmax out, [4] / [5]
    wave-level context evaluations / fall-backs so far IN THE SAMPLE (every 8th workgroup counts), [6] launches, [7] launches that sat out. The words only steer which of two exact
    softmax paths runs, so launches of several streams may share them. Never created while a stream is being captured (the zero-fill
    would become a node of the graph and wipe the words at every replay): a capture without a buffer runs without statistics."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    t = _PROJ_STATS.get(device)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        t = _PROJ_STATS[device] = torch.zeros(_lib.P3_STATS_WORDS, dtype=torch.int32, device=device)
    return t


def proj_stats_summary():
    """{evaluations, fallbacks (both in the 1-in-8 workgroup sample), rate, launches, launches_sat_out, sitting_out_now} over every
    statistics buffer of the process."""
    tot = [0] * _lib.P3_STATS_WORDS
    for t in _PROJ_STATS.values():
        for i, v in enumerate(t.cpu().tolist()):
            tot[i] += v & 0xffffffff
    return {"sampled_evaluations": tot[4], "sampled_fallbacks": tot[5], "sample": "every 8th workgroup", "fallback_rate": (tot[5] / tot[4]) if tot[4] else 0.0, "launches": tot[6],
            "launches_sat_out": tot[7], "sitting_out_now": tot[0]}


class _XAttnBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, coef, packed, mask, scale):
        out, _ = xattn_forward(q, packed, mask, coef, scale)
        ctx.packed, ctx.mask, ctx.scale = packed, mask, scale
        ctx.save_for_backward(q, coef if coef is not None else q.new_empty(0))
        ctx.has_coef = coef is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        q, coef = ctx.saved_tensors
        coef_in = coef if ctx.has_coef else None
        dq, dcoef = xattn_backward(q, ctx.packed, ctx.mask, coef_in, dout, ctx.scale)
        if ctx.has_coef:
            dcoef = dcoef.to(coef.dtype).reshape(coef.shape)
        else:
            dcoef = None
        return dq, dcoef, None, None, None


def xattn_blend(q, coef, packed, mask, scale):
    """Differentiable fused op: q [2I,N,C], coef [K] or [I,K] -> blended pre-projection output [2I,N,C].

    Per image: out[0] = A(q[0]; ctx 0);  out[1] = A(q[1]; ctx 1) + sum_i coef[i] mask[i] (A(q[1]; ctx 2+i) - out[0]).
    Gradients flow to q and coef only (see include/sta_xattn.h).
    """
    return _XAttnBlend.apply(q, coef, packed, mask, scale)


LN2 = 0.6931471805599453     # pass as `scale` when q is already multiplied by (softmax scale * log2 e): the kernel's log2-domain fast path


def sfrag_channels():
    """[10, 4, 8] long tensor: channel behind slot j of lane row g of fragment f in the SELF-attention kernel's out-fragment order
    (csrc/sta_rowgemm.hip::sfrag_channel restated; C = 320, 8 heads of 40)."""
    t = torch.empty(10, 4, 8, dtype=torch.long)
    for f in range(10):
        for g in range(4):
            for j in range(8):
                t[f, g, j] = f * 40 + (4 * g + j if j < 4 else 16 + 4 * g + (j - 4)) if f < 8 else (4 * (f - 8) + g) * 40 + 32 + j
    return t


def from_sfrag(of):
    """Self-attention out-fragment order -> row-major [B, N, 320]; tests and tools only."""
    rows = of.numel() // 320
    t = of.reshape(rows // 16, 10, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(rows // 16, 16, 320)
    out = torch.empty_like(t)
    out[..., sfrag_channels().reshape(-1).to(of.device)] = t
    return out.reshape(of.shape)


def to_sfrag(x):
    rows = x.numel() // 320
    t = x.reshape(rows // 16, 16, 320)[..., sfrag_channels().reshape(-1).to(x.device)].reshape(rows // 16, 16, 10, 4, 8).permute(0, 2, 3, 1, 4)
    return t.contiguous().view(x.shape)


def reset_selfattn_optimistic_state():
    """Forget the optimistic self-attention kernel's per-(device, stream, shape) state words (calls still sitting the optimistic loop out):
    the next call of every shape starts optimistic again. Results never depend on the state — it only selects between two exact loops."""
    _SA_FLAGS.clear()


SELFATTN_OPTIMISTIC = os.environ.get("STA_SELFATTN_OPTIMISTIC", "1") != "0"      # (env: A/B in tools/sa_opt_bench_ab.sh) self-attention at level 0 (both 16-bit types) through sta_selfattn_fwd_optimistic (profiles/r05_selfattn.md)
_SA_FLAGS = {}


def self_attention_sfrag_supported(x, heads):
    B, N, C = x.shape
    return self_attention_supported(x, heads) and C == 320 and heads == 8 and N % 16 == 0


def self_attention(q, k, vt, heads, scale, sfrag=False):
    """Flash-style self-attention through the HIP kernel (inference only, no autograd).
    q, k: [B, N, C] (last dim contiguous; a row stride > C is allowed, e.g. slices of a fused QKV buffer);
    vt: [B, C, N] (V transposed), last dim contiguous, any batch / channel strides that are multiples of 8 — e.g.
    the [C, B*N] result of ONE GEMM W_v . X^T viewed as [B, C, N]. Returns [B, N, C]; `sfrag`: in the kernel's out-fragment order
    (C = 320, 8 heads, N % 16 == 0) for sta.fused.to_out_add_layernorm_ofrag with a FRAG_SELFATTN weight."""
    B, N, C = q.shape
    if k.shape != q.shape or tuple(vt.shape) != (B, C, N):
        raise ValueError("shapes: q/k [B,N,C], vt [B,C,N]; got %s %s %s" % (tuple(q.shape), tuple(k.shape), tuple(vt.shape)))
    if not q.is_cuda:
        raise RuntimeError("self_attention needs CUDA/HIP tensors (there is no CPU path)")
    for t in (q, k):
        if t.stride(2) != 1 or t.stride(0) != N * t.stride(1):
            raise ValueError("q/k must be row-major with a uniform row stride")
    if vt.stride(2) != 1 or vt.stride(0) % 8 or vt.stride(1) % 8 or vt.stride(1) < N:
        vt = vt.contiguous()
    out = torch.empty((B, N, C), dtype=q.dtype, device=q.device)
    L = _lib.load()
    if SELFATTN_OPTIMISTIC and L.sta_selfattn_optimistic_supported(N, C, heads, float(scale), _dtype_code(q)):
        # the pipelined kernel's shapes: key loop from the workgroup's own block, no running maximum behind a tile's first key block + a repair launch for the workgroups whose
        # denominators left the safe range (sta_selfattn_fwd_optimistic); one flag word per workgroup, cached per device
        nbytes = L.sta_selfattn_optimistic_flags_bytes(B, N, heads)
        key = (q.device, _stream(q), nbytes)     # per stream: two concurrent calls must not share flag words or the two state words
        flags = _SA_FLAGS.get(key)
        if flags is None:
            flags = _SA_FLAGS[key] = torch.zeros(nbytes, dtype=torch.uint8, device=q.device)      # zero once: the leading state words live across calls
        _lib.check(L.sta_selfattn_fwd_optimistic(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), flags.data_ptr(), B, N, C, heads,
                                                 q.stride(1), k.stride(1), vt.stride(1), vt.stride(0), float(scale), _dtype_code(q), 1 if sfrag else 0,
                                                 _stream(q)), "sta_selfattn_fwd_optimistic")
        return out
    fn, name = (L.sta_selfattn_fwd_sfrag, "sta_selfattn_fwd_sfrag") if sfrag else (L.sta_selfattn_fwd, "sta_selfattn_fwd")
    _lib.check(fn(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), B, N, C, heads,
                  q.stride(1), k.stride(1), vt.stride(1), vt.stride(0), float(scale), _dtype_code(q), _stream(q)), name)
    return out


def self_attention_lse(q, k, vt, heads, scale):
    """`self_attention` for the differentiable path: also returns lse [B, heads, N] float32 (log2 domain), the only
    thing the backward kernels need of the N x N scores (sta_selfattn_fwd_lse)."""
    B, N, C = q.shape
    out = torch.empty((B, N, C), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, heads, N), dtype=torch.float32, device=q.device)
    L = _lib.load()
    _lib.check(L.sta_selfattn_fwd_lse(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, C, heads,
                                      q.stride(1), k.stride(1), vt.stride(1), vt.stride(0), float(scale), _dtype_code(q), _stream(q)),
               "sta_selfattn_fwd_lse")
    return out, lse


class SelfAttentionQKV(torch.autograd.Function):
    """Differentiable self-attention over ONE fused projection buffer qkv [B, N, 3C] (columns [q | k | v]): the HIP
    forward keeps out and lse, the HIP backward (sta_selfattn_bwd: delta, dk/dv and dq kernels) writes the three column
    blocks of ONE [B, N, 3C] gradient, so the projection's backward is a single GEMM. Replaces the autograd of
    CrossAttention.forward with context = x (attention.py:175-197) in the tracked epochs (plms.py:275-277)."""

    @staticmethod
    def forward(ctx, qkv, heads, scale):
        B, N, C3 = qkv.shape
        C = C3 // 3
        if C3 != 3 * C or C % heads or (C // heads) % 8 or C // heads > 160 or N % 64 or not qkv.is_cuda or qkv.dtype not in _DTYPES:
            # refused here rather than at backward time (sta_selfattn_bwd streams whole 64-row blocks, d <= 160)
            raise ValueError("SelfAttentionQKV needs a CUDA 16-bit [B, N, 3C] buffer with N %% 64 == 0 and head dim %% 8 == 0, <= 160; "
                             "got %s heads=%d %s" % (tuple(qkv.shape), heads, qkv.dtype))
        qkv = qkv if qkv.is_contiguous() else qkv.contiguous()
        vt = qkv[..., 2 * C:].transpose(1, 2).contiguous()
        out, lse = self_attention_lse(qkv[..., :C], qkv[..., C:2 * C], vt, heads, scale)
        ctx.save_for_backward(qkv, out, lse)
        ctx.heads, ctx.scale = heads, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        B, N, C3 = qkv.shape
        C = C3 // 3
        dout = dout.contiguous()
        qt = qkv[..., :C].transpose(1, 2).contiguous()                 # [B, C, N] copies: the A operands whose MFMA k axis is pixels
        kt = qkv[..., C:2 * C].transpose(1, 2).contiguous()
        dot = dout.transpose(1, 2).contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        es = qkv.element_size()
        L = _lib.load()
        _lib.check(L.sta_selfattn_bwd(qkv.data_ptr(), qkv.data_ptr() + C * es, qkv.data_ptr() + 2 * C * es, qt.data_ptr(), kt.data_ptr(),
                                      dout.data_ptr(), dot.data_ptr(), out.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                      dqkv.data_ptr(), dqkv.data_ptr() + C * es, dqkv.data_ptr() + 2 * C * es,
                                      B, N, C, ctx.heads, 3 * C, 3 * C, float(ctx.scale), _dtype_code(qkv), _stream(qkv)),
                   "sta_selfattn_bwd")
        return dqkv, None, None


def self_attention_train_supported(x, heads):
    """Shapes the differentiable HIP self-attention takes (forward with lse + backward kernels; the rest stays on SDPA)."""
    if not SELFATTN_ENABLED:
        return False
    B, N, C = x.shape
    d = C // heads
    return x.is_cuda and x.dtype in _DTYPES and N % 64 == 0 and d % 8 == 0 and d <= 160 and C % heads == 0


def self_attention_supported(x, heads):
    """Shapes the HIP self-attention kernel takes (the rest goes to PyTorch's SDPA)."""
    from . import fused as _fused
    if not SELFATTN_ENABLED or _fused._hold:
        return False
    B, N, C = x.shape
    d = C // heads
    return x.is_cuda and x.dtype in _DTYPES and N % 8 == 0 and N >= 64 and d % 8 == 0 and d <= 160 and C % heads == 0
