"""GPU parity of the HIP kernels (through the C-ABI) against the CPU oracle on the same inputs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import xattn_oracle as orc  # noqa: E402


def _case(N, C, heads, K, dtype, seed=0, M=77, centres=None, dev="cuda"):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(2, N, C, generator=g)
    k = torch.randn(K + 2, M, C, generator=g) * 0.7
    v = torch.randn(K + 2, M, C, generator=g)
    dim = int(math.isqrt(N))
    if dim * dim == N and K > 0:
        centres = centres or [(0.30, 0.40), (0.70, 0.60), (0.5, 0.2), (0.25, 0.75), (0.05, 0.95), (0.9, 0.1), (0.5, 0.5), (0.0, 0.0)][:K]
        mask = orc.disc_masks(centres, dim)
    else:
        mask = torch.rand(K, N, generator=g) < 0.3
    coef = torch.rand(K, generator=g) * 3 + 0.5
    q, k, v = (t.to(dtype) for t in (q, k, v))
    return q, k, v, mask, coef


def _run_fwd(q, k, v, mask, coef, heads, want_maps):
    from sta import ops
    dev = "cuda"
    packed = ops.pack_kv(k.to(dev), v.to(dev), heads)
    scale = (q.shape[-1] // heads) ** -0.5
    out, maps = ops.xattn_forward(q.to(dev), packed, ops.mask_bits(mask).to(dev), coef.to(dev), scale, want_maps)
    torch.cuda.synchronize()
    return out.float().cpu(), None if maps is None else maps.cpu()


SHAPES = [
    # N, C, heads, K
    (64, 64, 8, 1),      # d = 8
    (256, 320, 8, 2),    # d = 40  (level 0 head dim)
    (64, 640, 8, 2),     # d = 80
    (64, 1280, 8, 2),    # d = 160
    (1024, 320, 8, 2),   # 2 waves / WG
    (4096, 320, 8, 2),   # BASELINE level-0 shape, 4 waves / WG
    (144, 640, 8, 4),    # 768^2 mid-ish N not multiple of 64, K = 4
    (100, 128, 4, 3),    # ragged N (not multiple of 16), d = 32
    (256, 192, 8, 0),    # no objects
    (9216, 320, 8, 4),   # BASELINE configs[4]: 768^2 level 0, 4 objects (staged kernel, 6 contexts in LDS)
    (2304, 640, 8, 4),   # 768^2 level 1
    (576, 1280, 8, 4),   # 768^2 level 2
    (1024, 320, 8, 8),   # K = 8 (maximum): wave w attends contexts w, w+4, w+8
    (4096, 384, 8, 6),   # d = 48, 8 contexts: the staged kernel needs two LDS groups
]


@pytest.mark.parametrize("N,C,heads,K", SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fwd_matches_oracle(N, C, heads, K, dtype):
    q, k, v, mask, coef = _case(N, C, heads, K, dtype)
    scale = (C // heads) ** -0.5
    ref, ref_maps = orc.fused_xattn(q.double(), k.double(), v.double(), mask, coef.double(), heads, scale, want_maps=True)
    for want_maps in (True, False):
        out, maps = _run_fwd(q, k, v, mask, coef, heads, want_maps)
        # outputs are rounded to bf16/fp16 and P is rounded to 16 bits before P.V
        eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        tol = 4 * eps * (1.0 + ref.abs())
        err = (out.double() - ref).abs()
        assert (err <= tol).all(), "max err %.4g (tol %.4g)" % (err.max(), tol.max())
        if want_maps:
            # north_star: per-step attention maps within 1e-3 of the CPU reference
            merr = (maps.double() - ref_maps).abs().max().item()
            assert merr < 1e-4, merr


@pytest.mark.parametrize("N,C,heads,K", SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_bwd_matches_oracle(N, C, heads, K, dtype):
    from sta import ops
    q, k, v, mask, coef = _case(N, C, heads, K, dtype, seed=1)
    scale = (C // heads) ** -0.5
    g = torch.Generator().manual_seed(7)
    dout = torch.randn(2, N, C, generator=g).to(dtype)
    qd = q.double().requires_grad_(True)
    cd = coef.double().requires_grad_(True)
    ref = orc.fused_xattn(qd, k.double(), v.double(), mask, cd, heads, scale)
    ref.backward(dout.double())
    dev = "cuda"
    packed = ops.pack_kv(k.to(dev), v.to(dev), heads)
    dq, dcoef = ops.xattn_backward(q.to(dev), packed, ops.mask_bits(mask).to(dev), coef.to(dev), dout.to(dev), scale)
    torch.cuda.synchronize()
    dq, dcoef = dq.float().cpu().double(), dcoef.cpu().double()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    # dS is rounded to 16 bits before the dS.K product: error scales with the gradient magnitude
    gscale = qd.grad.abs().max().item()
    err = (dq - qd.grad).abs().max().item()
    assert err <= 6 * eps * gscale + 1e-6, (err, gscale)
    if K:
        # dcoef_i sums N*C signed products of 16-bit-rounded factors: 2 % relative + 0.5 % of the largest component
        g = cd.grad
        tol = 0.02 * g.abs() + 0.005 * g.abs().max() + 1e-4 * math.sqrt(N * C)
        assert ((dcoef - g).abs() <= tol).all(), (dcoef, g)


@pytest.mark.parametrize("N,C,heads,K,I,tiles,slots,M", [
    (4096, 320, 8, 2, 1, None, 0, 77),    # level 0, one image: all four contexts resident (29 KiB each)
    (4096, 320, 8, 2, 3, None, 0, 77),    # 3 images: ragged tile count per workgroup
    (256, 320, 8, 2, 2, 2, 0, 77),        # small launch, forced tile count
    (1000, 160, 4, 1, 3, 3, 0, 77),       # ragged N, 4 heads, forced tile count
    (1024, 384, 8, 2, 2, None, 0, 77),    # d = 48
    (1024, 320, 8, 0, 2, None, 0, 77),    # no objects
    (1024, 640, 8, 2, 4, None, 0, 77),    # level 1 (d = 80): contexts 0, 1 and the first local one in LDS, the second from L2
    (1024, 640, 8, 4, 2, 3, 0, 77),       # d = 80, K = 4: three local contexts from L2
    (256, 1280, 8, 2, 4, None, 0, 77),    # level 2 (d = 160): both local contexts from L2
    (64, 1280, 8, 2, 8, None, 0, 77),     # middle block
    (576, 1280, 8, 4, 2, None, 0, 77),    # 768^2 level 2, K = 4
    (4096, 320, 8, 2, 2, 5, 2, 77),       # d = 40 with only two LDS slots: the L2 path on the level-0 shape
    (9216, 320, 8, 4, 1, None, 0, 77),    # 768^2 level 0, K = 4: six contexts, five fit
    (1024, 320, 8, 8, 1, None, 0, 77),    # K = 8 (maximum)
    (256, 320, 8, 2, 2, None, 0, 33),     # M <= 64: whole key tiles are padding (the predicate softmax)
    (100, 128, 4, 3, 2, None, 3, 16),     # ragged N (not a multiple of 16), d = 32, M = 16
    (4000, 896, 8, 2, 1, None, 0, 80),    # d = 112, M = 80, N not a multiple of the tile
    (4096, 320, 8, 2, 16, None, 0, 77),   # the tracked epochs' launches (16 prompts per step): level 0, 16 tiles per workgroup
    (1024, 640, 8, 2, 16, None, 0, 77),   # level 1
])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_bwd_lds_resident_matches_oracle(N, C, heads, K, I, tiles, slots, M, dtype):
    """The LDS-resident multi-tile backward (backward operand images of as many contexts as fit in LDS, the other local ones
    from L2, no barrier per context, no attention outputs formed) against the fp64 oracle per image, in every launch geometry:
    automatic, a forced tile count, a forced slot count (dq of two geometries of one launch: identical bits, every pixel sees the
    same arithmetic; dcoef: equal up to the order of the per-wave sums)."""
    from sta import lib, ops
    dev = "cuda"
    cases = [_case(N, C, heads, K, dtype, seed=50 + i, M=M) for i in range(I)]
    q = torch.cat([c[0] for c in cases]).to(dev); k = torch.cat([c[1] for c in cases]).to(dev); v = torch.cat([c[2] for c in cases]).to(dev)
    mb = torch.stack([ops.mask_bits(c[3]) for c in cases]).to(dev)
    coef = torch.stack([c[4] for c in cases]).to(dev)
    g = torch.Generator().manual_seed(17)
    dout = torch.randn(2 * I, N, C, generator=g).to(dtype).to(dev)
    scale = (C // heads) ** -0.5
    packed = ops.pack_kv(k, v, heads, n_img=I)
    try:
        lib.set_option(lib.OPT_STAGED_TILES, tiles or 0)
        lib.set_option(lib.OPT_BWD_SLOTS, slots)
        dq, dcoef = ops.xattn_backward(q, packed, mb, coef, dout, scale)
        lib.set_option(lib.OPT_STAGED_TILES, 1)
        lib.set_option(lib.OPT_BWD_SLOTS, 2)
        dq1, dcoef1 = ops.xattn_backward(q, packed, mb, coef, dout, scale)
        torch.cuda.synchronize()
    finally:
        for o in (lib.OPT_STAGED_TILES, lib.OPT_BWD_SLOTS):
            lib.set_option(o, 0)
    assert torch.equal(dq, dq1)
    if K:
        assert torch.allclose(dcoef, dcoef1, rtol=1e-4, atol=1e-3 * dcoef1.abs().max().item() + 1e-5)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for i in sorted({0, I - 1}):
        qi, ki, vi, mi, ci = cases[i]
        qd, cd = qi.double().requires_grad_(True), ci.double().requires_grad_(True)
        orc.fused_xattn(qd, ki.double(), vi.double(), mi, cd, heads, scale).backward(dout[2 * i:2 * i + 2].cpu().double())
        gs = qd.grad.abs().max().item()
        assert (dq[2 * i:2 * i + 2].float().cpu().double() - qd.grad).abs().max() <= 6 * eps * gs + 1e-6
        if K:
            gc = cd.grad
            tol = 0.02 * gc.abs() + 0.005 * gc.abs().max() + 1e-4 * math.sqrt(N * C)
            assert ((dcoef.view(I, K)[i].cpu().double() - gc).abs() <= tol).all()


@pytest.mark.parametrize("N,C,heads,K,I", [(9216, 320, 8, 4, 1), (9216, 320, 8, 2, 1), (4096, 320, 8, 2, 16), (1024, 640, 8, 2, 16), (256, 1280, 8, 2, 16)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_bwd_is_bit_reproducible(N, C, heads, K, I, dtype):
    """Eight launches of one backward problem give identical bits (dq and dcoef). The eight-wave build of this kernel did not, at
    d = 40 with several tiles per workgroup (a handful of row-0 pixels inside the discs, different ones every run:
    profiles/r06_bwd_race.md) — which is why the product runs four waves x one tile, and why this test exists."""
    from sta import ops
    dev = "cuda"
    cases = [_case(N, C, heads, K, dtype, seed=1 + i) for i in range(min(I, 2))]
    rep = lambda ts: torch.cat([ts[i % len(ts)] for i in range(I)])
    q, k, v = rep([c[0] for c in cases]).to(dev), rep([c[1] for c in cases]).to(dev), rep([c[2] for c in cases]).to(dev)
    mb = torch.stack([ops.mask_bits(cases[i % len(cases)][3]) for i in range(I)]).to(dev)
    coef = torch.stack([cases[i % len(cases)][4] for i in range(I)]).to(dev)
    g = torch.Generator().manual_seed(7)
    dout = torch.randn(2 * I, N, C, generator=g).to(dtype).to(dev)
    packed = ops.pack_kv(k, v, heads, n_img=I)
    scale = (C // heads) ** -0.5
    outs = [ops.xattn_backward(q, packed, mb, coef, dout, scale) for _ in range(8)]
    torch.cuda.synchronize()
    for dq, dc in outs[1:]:
        assert torch.equal(dq, outs[0][0]) and torch.equal(dc, outs[0][1])


@pytest.mark.parametrize("N,C,heads,K,M", [(64, 320, 8, 2, 80), (256, 320, 8, 2, 33), (16, 64, 8, 1, 77), (1024, 640, 8, 3, 1), (48, 1280, 8, 2, 16)])
def test_key_count_and_tiny_latents(N, C, heads, K, M):
    """Edge cases of the key axis (M = 1 .. 80 = STA_MAX_KEYS; CLIP uses 77) and of the pixel axis (a single
    16-pixel tile), forward and backward, against the oracle."""
    from sta import ops
    dtype, dev = torch.bfloat16, "cuda"
    q, k, v, mask, coef = _case(N, C, heads, K, dtype, seed=9, M=M)
    scale = (C // heads) ** -0.5
    qd, cd = q.double().requires_grad_(True), coef.double().requires_grad_(True)
    ref, ref_maps = orc.fused_xattn(qd, k.double(), v.double(), mask, cd, heads, scale, want_maps=True)
    packed = ops.pack_kv(k.to(dev), v.to(dev), heads)
    mb = ops.mask_bits(mask).to(dev)
    out, maps = ops.xattn_forward(q.to(dev), packed, mb, coef.to(dev), scale, want_maps=True)
    eps = 2.0 ** -8
    assert ((out.float().cpu().double() - ref.detach()).abs() <= 4 * eps * (1.0 + ref.detach().abs())).all()
    assert (maps.cpu().double() - ref_maps.detach()).abs().max() < 1e-4
    g = torch.Generator().manual_seed(11)
    dout = torch.randn(2, N, C, generator=g).to(dtype)
    ref.backward(dout.double())
    dq, dcoef = ops.xattn_backward(q.to(dev), packed, mb, coef.to(dev), dout.to(dev), scale)
    # (M = 1: the gradient is exactly 0; the kernel's P = exp2(fma(s, c, -max * c)) / sum is 1 up to one fp32 rounding of the exponent: |dq| < 1e-5)
    assert (dq.float().cpu().double() - qd.grad).abs().max() <= 6 * eps * qd.grad.abs().max() + 1e-5
    gc = cd.grad
    assert ((dcoef.cpu().double() - gc).abs() <= 0.02 * gc.abs() + 0.005 * gc.abs().max() + 1e-4 * math.sqrt(N * C)).all()
    with pytest.raises(RuntimeError, match="keys unsupported"):
        ops.pack_kv(torch.zeros(2, 81, C, device=dev, dtype=dtype), torch.zeros(2, 81, C, device=dev, dtype=dtype), heads)


def test_domain_properties_full_size():
    """Size-independent properties at the BASELINE level-0 shape (N=4096, C=320, K=2):
    (1) coef = 0 or empty discs -> plain attention of both rows; (2) out[1] is affine in coef;
    (3) swapping the two objects (contexts, mask bits, weights together) changes nothing;
    (4) out[0] does not depend on masks/weights at all."""
    from sta import ops
    N, C, heads, K = 4096, 320, 8, 2
    q, k, v, mask, coef = _case(N, C, heads, K, torch.bfloat16, seed=5)
    dev, scale = "cuda", (C // heads) ** -0.5
    packed = ops.pack_kv(k.to(dev), v.to(dev), heads)
    mb = ops.mask_bits(mask).to(dev)
    run = lambda pk, m, c: ops.xattn_forward(q.to(dev), pk, m, c.to(dev), scale)[0].float()
    base = run(packed, mb, coef)
    plain = ops.xattn_forward(q.to(dev), ops.pack_kv(k[:2].to(dev), v[:2].to(dev), heads), None, None, scale)[0].float()
    assert torch.equal(run(packed, mb, torch.zeros(K)), plain)                       # (1) zero weights
    assert torch.equal(run(packed, torch.zeros_like(mb), coef), plain)                # (1) empty discs
    assert torch.equal(base[0], plain[0])                                             # (4)
    o1, o2, o3 = run(packed, mb, coef), run(packed, mb, 2 * coef), run(packed, mb, 3 * coef)
    assert ((o3 - o2) - (o2 - o1)).abs().max() <= 0.02 * (o2 - o1).abs().max() + 2 ** -6      # (2) up to output rounding
    perm = [0, 1, 3, 2]
    packed_sw = ops.pack_kv(k[perm].to(dev), v[perm].to(dev), heads)
    mb_sw = ops.mask_bits(mask[[1, 0]]).to(dev)
    sw = run(packed_sw, mb_sw, coef[[1, 0]])
    assert (sw - base).abs().max() <= 2 ** -6 * (1 + base.abs().max())                # (3) up to summation order


@pytest.mark.parametrize("N,C,heads,K", [(4096, 320, 8, 2), (1024, 640, 8, 2), (64, 1280, 8, 2), (100, 128, 4, 3), (256, 192, 8, 0)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_batched_images_match_oracle_per_image(N, C, heads, K, dtype):
    """n_img = 3 images in one launch (forward and backward) == the oracle on each image."""
    from sta import ops
    I, dev = 3, "cuda"
    cases = [_case(N, C, heads, K, dtype, seed=20 + i) for i in range(I)]
    q = torch.cat([c[0] for c in cases]); k = torch.cat([c[1] for c in cases]); v = torch.cat([c[2] for c in cases])
    masks = [c[3] for c in cases]; coef = torch.stack([c[4] for c in cases]) if K else None
    scale = (C // heads) ** -0.5
    packed = ops.pack_kv(k.to(dev), v.to(dev), heads, n_img=I)
    mb = torch.stack([ops.mask_bits(m) for m in masks]).to(dev) if K else None
    out, maps = ops.xattn_forward(q.to(dev), packed, mb, coef.to(dev) if K else None, scale, want_maps=True)
    g = torch.Generator().manual_seed(3)
    dout = torch.randn(2 * I, N, C, generator=g).to(dtype)
    dq, dcoef = ops.xattn_backward(q.to(dev), packed, mb, coef.to(dev) if K else None, dout.to(dev), scale)
    torch.cuda.synchronize()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for i in range(I):
        qi, ki, vi, mi, ci = cases[i]
        qd = qi.double().requires_grad_(True)
        cd = ci.double().requires_grad_(True)
        ref, ref_maps = orc.fused_xattn(qd, ki.double(), vi.double(), mi, cd, heads, scale, want_maps=True)
        err = (out[2 * i:2 * i + 2].float().cpu().double() - ref.detach()).abs()
        assert (err <= 4 * eps * (1.0 + ref.detach().abs())).all(), (i, err.max())
        assert (maps[i].cpu().double() - ref_maps.detach()).abs().max() < 1e-4
        ref.backward(dout[2 * i:2 * i + 2].double())
        gs = qd.grad.abs().max().item()
        assert (dq[2 * i:2 * i + 2].float().cpu().double() - qd.grad).abs().max() <= 6 * eps * gs + 1e-6
        if K:
            gc = cd.grad
            tol = 0.02 * gc.abs() + 0.005 * gc.abs().max() + 1e-4 * math.sqrt(N * C)
            assert ((dcoef.view(I, K)[i].cpu().double() - gc).abs() <= tol).all()


@pytest.mark.parametrize("N,C,heads,K,I,iters", [
    (4096, 320, 8, 2, 16, None),   # bench shape (16 prompts per step): 12-wave workgroups walking 11 strided tiles each
    (4096, 320, 8, 2, 8, None),    # 8 prompts per step: 6 tiles each
    (1024, 640, 8, 2, 8, None),    # 8-wave workgroups, 2 tiles each
    (4096, 320, 8, 2, 4, None),    # ragged tile count per workgroup (22 tiles of 192 pixels over 8 workgroups)
    (4000, 320, 8, 3, 4, 5),       # N not a multiple of the tile, forced tile count
    (4096, 384, 8, 6, 4, 3),       # 8 contexts do not fit LDS together: groups are re-staged for every tile
    (256, 1280, 8, 2, 8, None),    # d = 160: two LDS groups per tile
    # the launches of the default bench step (bench.py: 32 prompts per UNet call, fp16) at the four SD-v1 levels
    (4096, 320, 8, 2, 32, None),   # level 0 through sta_xattn_fwd (the tracked epochs / K > 2 take it; fixed weights fuse to_q)
    (1024, 640, 8, 2, 32, None),   # level 1
    (256, 1280, 8, 2, 32, None),   # level 2
    (64, 1280, 8, 2, 32, None),    # middle block
    # locals from L2 (the K + 2 contexts of a head exceed a CU's LDS): level 1 at K = 4, d = 160 with forced tile counts / ragged N
    (1024, 640, 8, 4, 8, None),
    (256, 1280, 8, 3, 16, 2),
    (250, 1280, 8, 2, 8, None),
    (576, 1280, 8, 4, 8, None),    # level 2 of a 768 x 768 image, K = 4 (BASELINE configs[4])
])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_multi_tile_workgroups(N, C, heads, K, I, iters, dtype):
    """Throughput-regime launches (several images, workgroups that keep one head's fragments in LDS and walk
    several strided pixel tiles): every image equals (a) the same image launched alone through the
    wave-per-context kernel and (b), for the first and last image, the CPU oracle."""
    from sta import ops
    dev = "cuda"
    from sta import lib
    if iters is not None:
        lib.set_option(lib.OPT_STAGED_TILES, iters)
        lib.set_option(lib.OPT_FWD_KERNEL, lib.FWD_STAGED)
    cases = [_case(N, C, heads, K, dtype, seed=40 + i) for i in range(I)]
    q = torch.cat([c[0] for c in cases]).to(dev); k = torch.cat([c[1] for c in cases]).to(dev); v = torch.cat([c[2] for c in cases]).to(dev)
    mb = torch.stack([ops.mask_bits(c[3]) for c in cases]).to(dev)
    coef = torch.stack([c[4] for c in cases]).to(dev)
    scale = (C // heads) ** -0.5
    out, _ = ops.xattn_forward(q, ops.pack_kv(k, v, heads, n_img=I), mb, coef, scale)
    torch.cuda.synchronize()
    lib.set_option(lib.OPT_STAGED_TILES, 0)
    lib.set_option(lib.OPT_FWD_KERNEL, lib.FWD_SPLIT)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for i in (range(I) if I <= 16 else (0, 1, I // 2, I - 2, I - 1)):
        qi, ki, vi, mi, ci = cases[i]
        alone, _ = ops.xattn_forward(qi.to(dev), ops.pack_kv(ki.to(dev), vi.to(dev), heads), ops.mask_bits(mi).to(dev), ci.to(dev), scale)
        a, b = out[2 * i:2 * i + 2].float(), alone.float()
        assert ((a - b).abs() <= 2 * eps * (1.0 + b.abs())).all(), (i, (a - b).abs().max().item())
        if i in (0, I - 1):
            ref = orc.fused_xattn(qi.double(), ki.double(), vi.double(), mi, ci.double(), heads, scale)
            err = (a.cpu().double() - ref).abs()
            assert (err <= 4 * eps * (1.0 + ref.abs())).all(), (i, err.max().item())
    lib.set_option(lib.OPT_FWD_KERNEL, 0)


@pytest.mark.parametrize("N,C,heads,K,I,tiles", [(256, 1280, 8, 2, 32, 0), (256, 1280, 8, 2, 16, 1), (64, 1280, 8, 2, 32, 1), (1024, 640, 8, 4, 8, 0),
                                                   (576, 1280, 8, 4, 4, 0), (256, 1152, 8, 1, 8, 2), (256, 896, 8, 5, 8, 1)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fwd_locals_from_l2_equals_grouped_staging(N, C, heads, K, I, tiles, dtype):
    """Where a head's K + 2 contexts do not fit a CU's LDS (d > 64) and a workgroup walks at least two tiles, the LDS-resident forward keeps
    the two mandatory contexts and reads the local ones as MFMA operands from L2; STA_OPT_PROJ_LL2 = 2 keeps the round-1 variant (contexts
    staged in groups, re-staged per tile), which one-tile launches take anyway (`tiles` > 0 forces the L2 variant with that tile count there).
    Same arithmetic in the same order: identical bits; and the oracle on the first / last image."""
    from sta import lib, ops
    dev = "cuda"
    cases = [_case(N, C, heads, K, dtype, seed=300 + i) for i in range(I)]
    q = torch.cat([c[0] for c in cases]).to(dev); k = torch.cat([c[1] for c in cases]).to(dev); v = torch.cat([c[2] for c in cases]).to(dev)
    mb = torch.stack([ops.mask_bits(c[3]) for c in cases]).to(dev)
    coef = torch.stack([c[4] for c in cases]).to(dev)
    scale = (C // heads) ** -0.5
    packed = ops.pack_kv(k, v, heads, n_img=I)
    lib.set_option(lib.OPT_FWD_KERNEL, lib.FWD_STAGED)
    try:
        lib.set_option(lib.OPT_STAGED_TILES, tiles)
        out, _ = ops.xattn_forward(q, packed, mb, coef, scale)
        lib.set_option(lib.OPT_STAGED_TILES, 0)
        lib.set_option(lib.OPT_PROJ_LL2, 2)
        grouped, _ = ops.xattn_forward(q, packed, mb, coef, scale)
        torch.cuda.synchronize()
    finally:
        lib.set_option(lib.OPT_STAGED_TILES, 0)
        lib.set_option(lib.OPT_PROJ_LL2, 0)
        lib.set_option(lib.OPT_FWD_KERNEL, 0)
    assert torch.equal(out, grouped)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for i in (0, I - 1):
        qi, ki, vi, mi, ci = cases[i]
        ref = orc.fused_xattn(qi.double(), ki.double(), vi.double(), mi, ci.double(), heads, scale)
        err = (out[2 * i:2 * i + 2].float().cpu().double() - ref).abs()
        assert (err <= 4 * eps * (1.0 + ref.abs())).all(), (i, err.max().item())


@pytest.mark.parametrize("B,N,C,heads", [(2, 4096, 320, 8), (4, 1024, 640, 8), (2, 144, 640, 8), (2, 576, 192, 4),
                                          (2, 64, 64, 8), (3, 200, 256, 4), (2, 2304, 640, 8),
                                          # d = 160 (SD-v1 levels 2 and mid at 512^2 / 768^2), 128, 112, 144
                                          (2, 256, 1280, 8), (4, 64, 1280, 8), (2, 576, 1280, 8), (2, 144, 1280, 8),
                                          (1, 128, 256, 2), (1, 72, 112, 1), (2, 192, 288, 2),
                                          (2, 1096, 80, 2), (1, 1024, 96, 2),       # d = 40 / 48 with a ragged / exact eight-wave tiling
                                          (3, 272, 320, 8)])                        # C = 320, N % 16 == 0 but not % 32: a half-empty last wave tile
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_self_attention_matches_reference(B, N, C, heads, dtype):
    """Flash-style self-attention kernel (attn1) vs softmax(q k^T scale) v in fp64 on the same 16-bit inputs;
    q/k taken as strided slices of one [B, N, 2C] buffer, V transposed — as the block feeds them."""
    from sta import ops
    g = torch.Generator().manual_seed(N + C)
    qk = torch.randn(B, N, 2 * C, generator=g).to(dtype)
    v = torch.randn(B, N, C, generator=g).to(dtype)
    d, scale = C // heads, (C // heads) ** -0.5
    qk_d, vt_d = qk.cuda(), v.transpose(1, 2).contiguous().cuda()
    out = ops.self_attention(qk_d[..., :C], qk_d[..., C:], vt_d, heads, scale)
    # the same V as the [C, B*N] result of ONE GEMM over the flattened batch, read through (row, batch) strides
    vt_flat = v.cuda().reshape(B * N, C).t().contiguous().view(C, B, N).permute(1, 0, 2)
    assert not vt_flat.is_contiguous() or B == 1
    out2 = ops.self_attention(qk_d[..., :C], qk_d[..., C:], vt_flat, heads, scale)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    q64 = qk[..., :C].double().view(B, N, heads, d).transpose(1, 2)
    k64 = qk[..., C:].double().view(B, N, heads, d).transpose(1, 2)
    v64 = v.double().view(B, N, heads, d).transpose(1, 2)
    ref = (torch.softmax(q64 @ k64.transpose(-1, -2) * scale, -1) @ v64).transpose(1, 2).reshape(B, N, C)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    err = (out.float().cpu().double() - ref).abs()
    assert (err <= 4 * eps * (1.0 + ref.abs())).all(), (err.max(), ref.abs().max())
    # log2-domain fast path (what the module runs: scale * log2 e folded into W_q): q' = round16(q * scale * log2 e), scale = ln 2;
    # the kernel then takes the exponent of the MFMA result directly. Reference: softmax of the SAME rounded q'.
    qs = (qk[..., :C].float() * (scale * 1.4426950408889634)).to(dtype)
    out3 = ops.self_attention(qs.cuda(), qk_d[..., C:], vt_d, heads, ops.LN2)
    torch.cuda.synchronize()
    qs64 = qs.double().view(B, N, heads, d).transpose(1, 2)
    ref3 = (torch.softmax(qs64 @ k64.transpose(-1, -2) * ops.LN2, -1) @ v64).transpose(1, 2).reshape(B, N, C)
    err3 = (out3.float().cpu().double() - ref3).abs()
    assert (err3 <= 4 * eps * (1.0 + ref3.abs())).all(), (err3.max(), ref3.abs().max())
    if ops.self_attention_sfrag_supported(qk_d[..., :C], heads):
        # the same launches leaving their output in the kernel's OUT-FRAGMENT order (for the fused to_out + residual + norm2 pass):
        # the same values at permuted addresses — up to the last bit of a few in fp16, where hipcc fuses `o * (1 / l)` with the
        # conversion in one epilogue (v_fma_mixlo_f16: one rounding) and not in the other (measured: 1e-5 of the values, 1 ulp)
        for q_, sc_, ref_, ref64 in ((qk_d[..., :C], scale, out, ref), (qs.cuda(), ops.LN2, out3, ref3)):
            o_f = ops.from_sfrag(ops.self_attention(q_, qk_d[..., C:], vt_d, heads, sc_, sfrag=True))
            torch.cuda.synchronize()
            assert (o_f != ref_).float().mean() < 1e-3 and ((o_f.float() - ref_.float()).abs() <= 2.0 ** -10 * ref_.float().abs() + 1e-12).all()
            e_f = (o_f.float().cpu().double() - ref64).abs()
            assert (e_f <= 4 * eps * (1.0 + ref64.abs())).all(), (e_f.max(), ref64.abs().max())
    elif C != 320:
        with pytest.raises(RuntimeError, match="C = 320"):
            ops.self_attention(qk_d[..., :C], qk_d[..., C:], vt_d, heads, scale, sfrag=True)
    if d <= 48:
        # both geometries of the log2-domain kernel at d <= 48: four waves x two query tiles (three waves per SIMD, shipped) and
        # eight waves x one tile (four per SIMD; opt-in through STA_OPT_SELFATTN_WAVES)
        from sta import lib
        for waves in (4, 8):
            lib.set_option(lib.OPT_SELFATTN_WAVES, waves)
            o_w = ops.self_attention(qs.cuda(), qk_d[..., C:], vt_d, heads, ops.LN2)
            torch.cuda.synchronize()
            e_w = (o_w.float().cpu().double() - ref3).abs()
            assert (e_w <= 4 * eps * (1.0 + ref3.abs())).all(), (waves, e_w.max(), ref3.abs().max())
        lib.set_option(lib.OPT_SELFATTN_WAVES, 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_self_attention_768_level0(dtype):
    """BASELINE configs[4] (768x768): attn1 of level 0 has N = 9216 queries and keys (d = 40). The full fp64 score tensor
    would be 11 GB, so 512 seeded query rows per (batch, head) are checked against softmax(q k^T scale) v in fp64 over
    ALL 9216 keys, plus the rows of the last (partial-free) tile."""
    from sta import ops
    B, N, C, heads = 2, 9216, 320, 8
    g = torch.Generator().manual_seed(9216)
    qk = torch.randn(B, N, 2 * C, generator=g).to(dtype)
    v = torch.randn(B, N, C, generator=g).to(dtype)
    d, scale = C // heads, (C // heads) ** -0.5
    out = ops.self_attention(qk[..., :C].cuda(), qk[..., C:].cuda(), v.transpose(1, 2).contiguous().cuda(), heads, scale)
    torch.cuda.synchronize()
    rows = torch.cat([torch.randperm(N, generator=g)[:512], torch.arange(N - 128, N)]).unique()
    q64 = qk[:, rows, :C].double().view(B, len(rows), heads, d).transpose(1, 2)
    k64 = qk[..., C:].double().view(B, N, heads, d).transpose(1, 2)
    v64 = v.double().view(B, N, heads, d).transpose(1, 2)
    ref = (torch.softmax(q64 @ k64.transpose(-1, -2) * scale, -1) @ v64).transpose(1, 2).reshape(B, len(rows), C)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    err = (out[:, rows.cuda()].float().cpu().double() - ref).abs()
    assert (err <= 4 * eps * (1.0 + ref.abs())).all(), (err.max(), ref.abs().max())


@pytest.mark.parametrize("pre", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_self_attention_deferred_rescale_branches(dtype, pre):
    """The running maximum of the online softmax is moved only when a pixel's maximum grew by more than 2^8 since the
    last move (sta_selfattn.hip). Inputs that force both sides of that branch at chosen key blocks: (a) keys whose norm
    ramps up along the sequence, so the maximum creeps up by less than the threshold per block but far more than it in
    total (the stale maximum must be refreshed several times); (b) spiked keys late in the sequence that lift the maximum
    of SOME queries of a wave by >> 2^8 in one block (all 16 pixels of the wave rescale, the others by a factor ~1);
    (c) a spike in the very last, partial block. Reference: fp64 softmax over all keys."""
    from sta import ops
    B, N, C, heads = 2, 1000, 320, 8            # N % 64 != 0: the last block is partial
    d, scale = C // heads, (C // heads) ** -0.5
    g = torch.Generator().manual_seed(123)
    q = torch.randn(B, N, C, generator=g)
    k = torch.randn(B, N, C, generator=g) * (0.3 + 2.7 * torch.arange(N).view(1, N, 1) / N)        # (a)
    for key, px in ((700, 5), (701, 260), (990, 17)):                                                 # (b), (c)
        k[:, key] = 4.0 * q[:, px]
    k[:, :64] = -1.5 * q[:, 40:41]              # (d) every score of pixel 40 in block 0 is far below zero: the first maximum is negative
    v = torch.randn(B, N, C, generator=g)
    if pre:                                    # the log2-domain kernels: q pre-multiplied by scale * log2 e, scale = ln 2
        q, scale = q * (scale * 1.4426950408889634), ops.LN2
    q, k, v = (t.to(dtype) for t in (q, k, v))
    out = ops.self_attention(q.cuda(), k.cuda(), v.transpose(1, 2).contiguous().cuda(), heads, scale)
    torch.cuda.synchronize()
    q64 = q.double().view(B, N, heads, d).transpose(1, 2)
    k64 = k.double().view(B, N, heads, d).transpose(1, 2)
    v64 = v.double().view(B, N, heads, d).transpose(1, 2)
    logits = q64 @ k64.transpose(-1, -2) * scale
    assert (logits.max(-1).values - logits[..., :64].max(-1).values).max() * 1.4427 > 40      # the maximum really moves late
    ref = (torch.softmax(logits, -1) @ v64).transpose(1, 2).reshape(B, N, C)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    err = (out.float().cpu().double() - ref).abs()
    assert torch.isfinite(out).all()
    assert (err <= 4 * eps * (1.0 + ref.abs())).all(), (err.max(), ref.abs().max())


@pytest.mark.parametrize("N", [64, 128, 192, 256, 320, 1024, 4096])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_self_attention_software_pipelined_loop(N, dtype):
    """The level-0 launch of attn1 (d = 40, 8 heads, q in log2 units, whole 64-key blocks) runs a software-pipelined loop — the two
    query tiles of a wave half a block apart, the MFMAs of one with the softmax of the other between them, operand registers
    double-buffered across the mid-block rendezvous, K as a k = 32 + a k = 16 operand per key tile (csrc/sta_selfattn.hip,
    selfattn_fwd_pipe_kernel). Its two geometries (128 and 256 queries per workgroup) must equal each other BIT FOR BIT and the plain
    loop to 2 eps (dims 32..39 enter through another MFMA shape), for 1, 2, 3, 4, 5 (prologue / drain / every ring slot) and many blocks,
    with keys that force the deferred-rescale branch of both tiles at chosen blocks, in row-major and out-fragment order, with and
    without the log-sum-exp; and the fp64 softmax to 4 eps."""
    from sta import lib, ops
    B, C, heads = 2, 320, 8
    d = C // heads
    g = torch.Generator().manual_seed(N)
    q = torch.randn(B, N, C, generator=g)
    k = torch.randn(B, N, C, generator=g) * (0.3 + 2.7 * torch.arange(N).view(1, N, 1) / N)      # the maximum creeps up along the sequence
    for key, px in ((N - 30, 5), (N // 2 + 1, 17 % N), (N - 1, 40)):                              # spikes: tile A and tile B pixels, late blocks
        k[:, key] = 4.0 * q[:, px]
    k[:, :64] = torch.where(torch.arange(64).view(1, 64, 1) < 32, -1.5 * q[:, 20:21], k[:, :64])   # pixel 20: half of block 0 far below zero
    v = torch.randn(B, N, C, generator=g)
    qs = (q * (d ** -0.5 * 1.4426950408889634)).to(dtype)
    k, v = k.to(dtype), v.to(dtype)
    qd, kd, vtd = qs.cuda(), k.cuda(), v.transpose(1, 2).contiguous().cuda()
    outs, opt = {}, {}
    try:
        for mode in (2, 4, 8, 3, 0):
            lib.set_option(lib.OPT_SELFATTN_PIPE, mode)
            ops.SELFATTN_OPTIMISTIC = False           # the standard loop: what the bit-for-bit claims are about
            o = ops.self_attention(qd, kd, vtd, heads, ops.LN2)
            o_f = ops.from_sfrag(ops.self_attention(qd, kd, vtd, heads, ops.LN2, sfrag=True)) if N % 16 == 0 else None
            o_l, lse = ops.self_attention_lse(qd, kd, vtd, heads, ops.LN2) if hasattr(ops, "self_attention_lse") else (o, None)
            ops.SELFATTN_OPTIMISTIC = True            # the product's call: optimistic loop from the workgroup's own key block + repair launch
            if mode != 2:
                ops._SA_FLAGS.clear()                 # a fresh flags buffer: no sitting out after a call with many failures
                opt[mode] = (ops.self_attention(qd, kd, vtd, heads, ops.LN2),
                             ops.from_sfrag(ops.self_attention(qd, kd, vtd, heads, ops.LN2, sfrag=True)) if N % 16 == 0 else None)
            torch.cuda.synchronize()
            outs[mode] = (o, o_f, o_l, lse)
    finally:
        ops.SELFATTN_OPTIMISTIC = True
        lib.set_option(lib.OPT_SELFATTN_PIPE, 0)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for mode in (8, 3, 0):                            # (8 waves / 3 tiles per wave forced at every N; the dispatcher's own choice is one of them)
        for a, b_ in zip(outs[mode], outs[4]):
            if a is not None:
                assert torch.equal(a, b_), (mode, (a.float() - b_.float()).abs().max().item())
    for a, b_ in zip(outs[4], outs[2]):
        if a is not None:
            assert ((a.float() - b_.float()).abs() <= 2 * eps * (1.0 + b_.float().abs())).all(), (a.float() - b_.float()).abs().max().item()
    for mode, pair in opt.items():                    # the optimistic call (its key loop starts at another block in every geometry): 2 eps of the standard loop
        for a, b_ in zip(pair, outs[4][:2]):
            if a is not None:
                assert torch.isfinite(a).all()
                assert ((a.float() - b_.float()).abs() <= 2 * eps * (1.0 + b_.float().abs())).all(), (mode, (a.float() - b_.float()).abs().max().item())
    q64 = qs.double().view(B, N, heads, d).transpose(1, 2)
    k64 = k.double().view(B, N, heads, d).transpose(1, 2)
    v64 = v.double().view(B, N, heads, d).transpose(1, 2)
    logits = q64 @ k64.transpose(-1, -2) * ops.LN2
    ref = (torch.softmax(logits, -1) @ v64).transpose(1, 2).reshape(B, N, C)
    for mode in (4, 8, 3):
        err = (outs[mode][0].float().cpu().double() - ref).abs()
        assert torch.isfinite(outs[mode][0]).all()
        assert (err <= 4 * eps * (1.0 + ref.abs())).all(), (mode, err.max(), ref.abs().max())
        err = (opt[mode][0].float().cpu().double() - ref).abs()
        assert (err <= 4 * eps * (1.0 + ref.abs())).all(), ("optimistic", mode, err.max(), ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("sfrag", [False, True])
def test_self_attention_optimistic_and_its_repair_launch(sfrag, dtype):
    """Level 0 takes sta_selfattn_fwd_optimistic: the key loop starts at the workgroup's own 64-key block, and behind a query tile's first
    block it keeps no running maximum (P = exp2(S - m_0): bf16 has fp32's exponent range, fp16 has 2^16 of headroom over m_0); every
    denominator is range-checked ([2^-100, 2^100) bf16, [2^-100, 2^15) fp16) and a flagged workgroup is recomputed by the standard loop in
    the same call. (a) ordinary logits — bf16: the maximum creeping up by ~40 log2 units along the sequence; fp16: by ~10 — no workgroup
    flagged, 4 eps of the fp64 softmax, 2 eps of the standard loop; (b) keys far from their queries that lift some queries' maximum by
    ~180 log2 units (fp16: ~20): exp2 overflows in the optimistic pass, their workgroups are flagged (others are not) and the result is exact again;
    (c) fp16 with the bf16 case's 40-unit creep: exact whatever is flagged."""
    from sta import lib, ops
    B, N, C, heads = 2, 1024, 320, 8
    d = C // heads
    bf = dtype == torch.bfloat16
    assert ops.SELFATTN_OPTIMISTIC and lib.load().sta_selfattn_optimistic_supported(N, C, heads, ops.LN2, lib.STA_BF16)
    assert lib.load().sta_selfattn_optimistic_supported(N, C, heads, ops.LN2, lib.STA_F16)
    g = torch.Generator().manual_seed(7)
    q = torch.randn(B, N, C, generator=g)
    ramp = torch.arange(N).view(1, N, 1) / N
    k0 = torch.randn(B, N, C, generator=g) * ((0.3 + 2.7 * ramp) if bf else (0.8 + 0.5 * ramp))
    k_steep = k0 * ((0.3 + 2.7 * ramp) / (0.8 + 0.5 * ramp))
    v = torch.randn(B, N, C, generator=g).to(dtype)
    qs = (q * (d ** -0.5 * 1.4426950408889634)).to(dtype)
    for case in ("ordinary", "spiked") + (() if bf else ("steep",)):
        k = (k_steep if case == "steep" else k0).clone()
        if case == "spiked":
            for key, px in ((700, 5), (990, 300)):          # batch 0 and 1, all heads: pixels 5 and 300 see a logit of several hundred at a late key
                # its own query: ~180 log2 units above everything else (fp16: ~27), other queries: +- 30 (1 sigma; fp16: +- 4.3)
                k[:, key] = (20.0 if bf else 3.0) * q[:, px]
        k = k.to(dtype)
        qd, kd, vtd = qs.cuda(), k.cuda(), v.transpose(1, 2).contiguous().cuda()
        fkey = (qd.device, torch.cuda.current_stream(qd.device).cuda_stream, lib.load().sta_selfattn_optimistic_flags_bytes(B, N, heads))
        assert fkey[2] == (B * heads * ((N + 127) // 128) + 32) * 4      # the two state words have a 128-byte line to themselves
        ops._SA_FLAGS[fkey] = torch.zeros(fkey[2], dtype=torch.uint8, device=qd.device)      # words beyond the chosen grid stay zero
        out = ops.self_attention(qd, kd, vtd, heads, ops.LN2, sfrag=sfrag)
        torch.cuda.synchronize()
        words = ops._SA_FLAGS[fkey].view(torch.int32).cpu()
        state, flags = words[:2].tolist(), words[32:]       # (calls to sit out, failures counted: reset), one flag word per workgroup of the grid the dispatcher chose
        assert state[1] == 0 and state[0] == (64 if 8 * int(flags.sum()) > 128 else 0), (state, int(flags.sum()))
        if state[0]:            # more than an eighth failed: the next call sits the optimistic loop out (all workgroups through the standard loop), exact all the same
            again = ops.self_attention(qd, kd, vtd, heads, ops.LN2, sfrag=sfrag)
            torch.cuda.synchronize()
            w2 = ops._SA_FLAGS[fkey].view(torch.int32).cpu()
            assert w2[:2].tolist() == [63, 0] and int(w2[32:32 + 128].sum()) == 128 and int(w2[2:32].sum()) == 0
            ops.SELFATTN_OPTIMISTIC = False
            try:
                assert torch.equal(again, ops.self_attention(qd, kd, vtd, heads, ops.LN2, sfrag=sfrag))
            finally:
                ops.SELFATTN_OPTIMISTIC = True
        out = ops.from_sfrag(out) if sfrag else out
        ops.SELFATTN_OPTIMISTIC = False
        try:
            std = ops.self_attention(qd, kd, vtd, heads, ops.LN2)
        finally:
            ops.SELFATTN_OPTIMISTIC = True
        q64 = qs.double().view(B, N, heads, d).transpose(1, 2)
        k64 = k.double().view(B, N, heads, d).transpose(1, 2)
        v64 = v.double().view(B, N, heads, d).transpose(1, 2)
        logits = q64 @ k64.transpose(-1, -2) * ops.LN2
        ref = (torch.softmax(logits, -1) @ v64).transpose(1, 2).reshape(B, N, C)
        eps = 2.0 ** -8 if bf else 2.0 ** -11
        err = (out.float().cpu().double() - ref).abs()
        assert torch.isfinite(out).all()
        assert (err <= 4 * eps * (1.0 + ref.abs())).all(), (case, err.max().item())
        assert ((out.float() - std.float()).abs() <= 2 * eps * (1.0 + std.float().abs())).all()
        # how far a row's maximum lies above the maximum over the query's own 64-key block (an upper bound of what the kernel's m_0, the
        # maximum over the WORKGROUP's first block at most three blocks away, has to absorb in these sequences without spatial structure)
        own = logits.view(B, heads, N // 64, 64, N // 64, 64).diagonal(dim1=2, dim2=4).amax(3).permute(0, 1, 3, 2).reshape(B, heads, N)
        grow = ((logits.max(-1).values - own) * 1.4426950408889634).max().item()
        if case == "ordinary":
            assert (15 < grow < 90 if bf else 2 < grow < 12) and int(flags.sum()) == 0, (grow, int(flags.sum()))
        elif case == "spiked":
            assert grow > (150 if bf else 17) and 0 < int(flags.sum()) < flags.numel(), (grow, int(flags.sum()), flags.numel())


SA_BWD_SHAPES = [(2, 256, 320, 8), (2, 128, 160, 2), (1, 64, 64, 4), (1, 192, 96, 1), (3, 320, 128, 2), (1, 4096, 320, 8), (1, 1024, 640, 8),
                 # d = 160 (levels 2 / mid at 512^2, level 2 at 768^2: single-buffered dk/dv kernel), 128, 112, 144
                 (2, 256, 1280, 8), (2, 64, 1280, 8), (1, 576, 1280, 8), (1, 128, 256, 2), (1, 64, 112, 1), (2, 192, 288, 2)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,N,C,heads", SA_BWD_SHAPES)
def test_self_attention_backward_matches_fp64(B, N, C, heads, dtype):
    """Differentiable self-attention (sta_selfattn_fwd_lse + sta_selfattn_bwd behind sta.ops.SelfAttentionQKV) against the
    fp64 autograd of softmax(q k^T scale) v on the same 16-bit [B, N, 3C] projection buffer (attention.py:175-197 with
    context = x): out, lse and the three column blocks of the gradient. Head dims 40 / 80 / 160 (the SD-v1 levels), 16 .. 144."""
    from sta import ops
    g = torch.Generator().manual_seed(N + C + B)
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dtype)
    dout = torch.randn(B, N, C, generator=g).to(dtype)
    d, scale = C // heads, (C // heads) ** -0.5
    x = qkv.cuda().requires_grad_(True)
    out = ops.SelfAttentionQKV.apply(x, heads, scale)
    out.backward(dout.cuda())
    _, lse = ops.self_attention_lse(x.detach()[..., :C], x.detach()[..., C:2 * C], x.detach()[..., 2 * C:].transpose(1, 2).contiguous(), heads, scale)
    torch.cuda.synchronize()
    dev64 = "cuda" if N > 1024 else "cpu"                 # the fp64 reference of the N = 4096 case runs on the GPU (PyTorch fp64 ops)
    r = qkv.double().to(dev64).requires_grad_(True)
    q64, k64, v64 = (r[..., i * C:(i + 1) * C].view(B, N, heads, d).transpose(1, 2) for i in range(3))
    logits = q64 @ k64.transpose(-1, -2) * scale
    ref = (torch.softmax(logits, -1) @ v64).transpose(1, 2).reshape(B, N, C)
    ref.backward(dout.double().to(dev64))
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    err = (out.detach().double().cpu() - ref.detach().cpu()).abs()
    assert (err <= 4 * eps * (1.0 + ref.detach().cpu().abs())).all(), err.max()
    lse_ref = torch.logsumexp(logits.detach(), -1).cpu() * 1.4426950408889634
    assert (lse.double().cpu() - lse_ref).abs().max() < 1e-3
    gref = r.grad.cpu()
    got = x.grad.double().cpu()
    for i, name in enumerate(("dq", "dk", "dv")):
        a, b_ = got[..., i * C:(i + 1) * C], gref[..., i * C:(i + 1) * C]
        # P and dS enter the gradient MFMAs rounded to 16 bits (as in every flash backward): the error of one output is a sum
        # of N rounded terms, bounded here relative to the largest gradient of the tensor
        assert (a - b_).abs().max() <= 6 * eps * b_.abs().max(), (name, (a - b_).abs().max(), b_.abs().max())
        assert ((a - b_).norm() / b_.norm()) < 2 * eps, (name, (a - b_).norm() / b_.norm())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,N,C,heads", [(1, 4096, 320, 8), (2, 256, 320, 8), (2, 1024, 640, 8), (2, 256, 1280, 8), (2, 64, 1280, 8), (3, 320, 320, 8)])
def test_self_attention_backward_log2_domain_q(B, N, C, heads, dtype):
    """What the model's tracked attn1 calls: scale = ln 2 on a q that carries d^-1/2 log2 e already (pre-scaled W_q) — the kernels whose
    S - lse and dP - delta come out of the MFMAs (accumulators start at -lse / -delta). Against the fp64 autograd of the same function of
    the same 16-bit buffer; eight- and four-wave workgroups give identical bits at d = 40."""
    from sta import lib, ops
    g = torch.Generator().manual_seed(B + N + C)
    d = C // heads
    qkv = torch.randn(B, N, 3 * C, generator=g)
    qkv[..., :C] *= d ** -0.5 * 1.4426950408889634          # q in log2 units
    qkv = qkv.to(dtype)
    dout = torch.randn(B, N, C, generator=g).to(dtype)
    x = qkv.cuda().requires_grad_(True)
    out = ops.SelfAttentionQKV.apply(x, heads, ops.LN2)
    out.backward(dout.cuda())
    got = x.grad.double().cpu()
    if d == 40 and N >= 1024:
        lib.set_option(lib.OPT_SELFATTN_WAVES, 4)
        try:
            x4 = qkv.cuda().requires_grad_(True)
            ops.SelfAttentionQKV.apply(x4, heads, ops.LN2).backward(dout.cuda())
        finally:
            lib.set_option(lib.OPT_SELFATTN_WAVES, 0)
        assert torch.equal(x4.grad, x.grad)
    dev64 = "cuda" if N > 1024 else "cpu"
    r = qkv.double().to(dev64).requires_grad_(True)
    q64, k64, v64 = (r[..., i * C:(i + 1) * C].view(B, N, heads, d).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q64 @ k64.transpose(-1, -2) * ops.LN2, -1) @ v64).transpose(1, 2).reshape(B, N, C)
    ref.backward(dout.double().to(dev64))
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    err = (out.detach().double().cpu() - ref.detach().cpu()).abs()
    assert (err <= 4 * eps * (1.0 + ref.detach().cpu().abs())).all(), err.max()
    gref = r.grad.cpu()
    for i, name in enumerate(("dq", "dk", "dv")):
        a, b_ = got[..., i * C:(i + 1) * C], gref[..., i * C:(i + 1) * C]
        assert (a - b_).abs().max() <= 6 * eps * b_.abs().max(), (name, (a - b_).abs().max(), b_.abs().max())
        assert ((a - b_).norm() / b_.norm()) < 2 * eps, (name, (a - b_).norm() / b_.norm())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_self_attention_backward_768_level0(dtype):
    """BASELINE configs[4] size (768x768: N = 9216 at level 0, d = 40): the HIP backward against fp32 autograd of the explicit
    softmax on the GPU (the fp64 form would need 22 GB per tensor), plus linearity in dout — a size-independent property of
    the backward: bwd(2 a - 3 b) = 2 bwd(a) - 3 bwd(b) up to the 16-bit rounding of the outputs."""
    from sta import ops
    B, N, C, heads = 1, 9216, 320, 8
    d, scale = C // heads, (C // heads) ** -0.5
    g = torch.Generator().manual_seed(77)
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dtype).cuda()
    da = torch.randn(B, N, C, generator=g).to(dtype).cuda()
    db = torch.randn(B, N, C, generator=g).to(dtype).cuda()

    def hip_grad(dout):
        x = qkv.clone().requires_grad_(True)
        ops.SelfAttentionQKV.apply(x, heads, scale).backward(dout)
        return x.grad.float()
    ga, gb = hip_grad(da), hip_grad(db)
    mix = (2.0 * da.float() - 3.0 * db.float()).to(dtype)
    gm = hip_grad(mix)
    r = qkv.float().requires_grad_(True)
    q, k, v = (r[..., i * C:(i + 1) * C].view(B, N, heads, d).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(B, N, C)
    ref.backward(da.float())
    torch.cuda.synchronize()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for i, name in enumerate(("dq", "dk", "dv")):
        a, b_ = ga[..., i * C:(i + 1) * C], r.grad[..., i * C:(i + 1) * C]
        assert (a - b_).abs().max() <= 6 * eps * b_.abs().max(), (name, (a - b_).abs().max(), b_.abs().max())
        assert (a - b_).norm() / b_.norm() < 2 * eps, name
    lin = 2.0 * ga - 3.0 * gb
    assert (gm - lin).norm() / lin.norm() < 4 * eps, (gm - lin).norm() / lin.norm()


def test_self_attention_tracked_module_matches_sdpa():
    """CrossAttention (attn1) with autograd enabled and frozen weights takes the HIP forward + backward; its input gradient
    and output against PyTorch SDPA autograd on the same module (sta.ops.SELFATTN_ENABLED off)."""
    from ldm.modules.attention import CrossAttention
    from sta import ops
    from sta.synth import seeded_fill_
    B, N, C, heads = 2, 1024, 320, 8
    attn = CrossAttention(query_dim=C, heads=heads, dim_head=C // heads)
    seeded_fill_(attn, 5)
    attn = attn.cuda().to(torch.float16).requires_grad_(False)
    x0 = torch.randn(B, N, C, generator=torch.Generator().manual_seed(1)).to(torch.float16).cuda()
    w = torch.randn(B, N, C, generator=torch.Generator().manual_seed(2)).to(torch.float16).cuda()
    res = []
    for enabled in (True, False):
        ops.SELFATTN_ENABLED = enabled
        try:
            x = x0.clone().requires_grad_(True)
            y = attn(x)
            (y.float() * w.float()).sum().backward()
            res.append((y.detach().float(), x.grad.float()))
        finally:
            ops.SELFATTN_ENABLED = True
    torch.cuda.synchronize()
    assert getattr(attn, "_wqkv", None) is not None          # the tracked HIP path ran
    for a, b_ in zip(res[0], res[1]):
        assert (a - b_).abs().max() <= 2.0 ** -7 * b_.abs().max(), ((a - b_).abs().max(), b_.abs().max())


def test_self_attention_module_large_batch():
    """CrossAttention's inference path at CFG batch 40 (20 prompts per UNet call): q/k from one fused GEMM, V^T from ONE
    plain GEMM over the flattened batch. (A weight-broadcast batched matmul for V^T faulted inside the GEMM library from
    batch 40 up — this pins the replacement.) Reference: PyTorch SDPA on the same projections."""
    from ldm.modules.attention import CrossAttention
    from sta.synth import seeded_fill_
    B, N, C, heads = 40, 1024, 640, 8
    attn = CrossAttention(query_dim=C, heads=heads, dim_head=C // heads)
    seeded_fill_(attn, 5)
    attn = attn.cuda().to(torch.bfloat16)
    x = torch.randn(B, N, C, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).cuda()
    with torch.no_grad():
        got = attn(x)
        q, k, v = (m(x).view(B, N, heads, -1).transpose(1, 2) for m in (attn.to_q, attn.to_k, attn.to_v))
        ref = attn.to_out(torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, C))
    torch.cuda.synchronize()
    err = (got.float() - ref.float()).abs()
    assert (err <= 2.0 ** -6 * (1.0 + ref.float().abs())).all(), err.max().item()


def test_autograd_function_roundtrip():
    from sta import ops
    N, C, heads, K = 256, 320, 8, 2
    q, k, v, mask, coef = _case(N, C, heads, K, torch.bfloat16, seed=3)
    dev = "cuda"
    packed = ops.pack_kv(k.to(dev), v.to(dev), heads)
    qg = q.to(dev).requires_grad_(True)
    W = torch.full((K, 50), 2.5, device=dev, requires_grad=True)   # plms.py:204-209 leaf
    out = ops.xattn_blend(qg, W[:, 7], packed, ops.mask_bits(mask).to(dev), (C // heads) ** -0.5)
    out.float().square().sum().backward()
    assert qg.grad is not None and qg.grad.shape == qg.shape
    assert W.grad is not None and W.grad[:, 7].abs().sum() > 0 and W.grad[:, :7].abs().sum() == 0


def test_error_convention():
    from sta import lib, ops
    L = lib.load()
    assert L.sta_version() == 0x000500
    assert L.sta_xattn_packed_kv_bytes(4, 8, 41) == 0          # d % 8 != 0
    x = torch.zeros(2, 16, 8 * 168, device="cuda", dtype=torch.bfloat16)
    rc = L.sta_xattn_fwd(x.data_ptr(), x.data_ptr(), 0, 0, x.data_ptr(), 0, 1, 16, 8 * 168, 8, 77, 0, 1.0, 0, 0)
    assert rc == -2 and "head dim" in lib.last_error()
    with pytest.raises(ValueError):
        ops.pack_kv(torch.zeros(2, 77, 8 * 168, device="cuda", dtype=torch.bfloat16),
                    torch.zeros(2, 77, 8 * 168, device="cuda", dtype=torch.bfloat16), 8)


# ---------------------------------------------------------------------------------------------------
# forward with the query projection inside (sta_xattn_fwd_proj)
# ---------------------------------------------------------------------------------------------------
PROJ_SHAPES = [
    # N, C, heads, K, images, forced tiles per workgroup (None = heuristic), waves per workgroup (0 = default),
    # pair (True: d = 40, C = 160 / 320 and K <= 2 take the head-PAIR kernel, sta_xattn_proj3.hip; False: one head per workgroup forced)
    (256, 320, 8, 2, 1, None, 0, True),      # level-0 head dim, two 128-pixel tiles
    (256, 320, 8, 2, 1, None, 0, False),
    (4096, 320, 8, 2, 4, None, 0, True),     # BASELINE level 0, 4 prompts
    (4096, 320, 8, 2, 16, None, 0, True),    # the bench launch: 16 prompts, 4 head pairs x 4 workgroups per image, 8 tiles each
    (4096, 320, 8, 2, 16, None, 0, False),   # 16 tiles per workgroup, one head each
    (4096, 320, 8, 2, 32, None, 0, True),    # the DEFAULT bench launch (32 prompts per UNet call): 256 pair workgroups x 16 tiles, one round
    (4096, 320, 8, 1, 3, None, 4, True),     # one object, 4-wave workgroups
    (4096, 320, 8, 0, 2, None, 0, True),     # no objects
    (1000, 160, 4, 2, 2, 3, 4, True),        # ragged N, 2 head pairs, forced tile count
    (4096, 320, 8, 4, 2, None, 0, True),     # 4 objects do not fit as pairs -> per-head kernel (6 contexts + Wq = 144 KiB)
    (1000, 160, 4, 3, 2, 3, 4, True),        # K = 3: per-head kernel, ragged N, forced tile count, 4-wave workgroups
    (1024, 320, 4, 1, 2, None, 0, True),     # d = 80: 50 KiB of Wq + 3 contexts (per head)
    (576, 480, 10, 0, 1, None, 0, True),     # d = 48 with 10 heads (15 k-steps), no objects (per head)
    (9216, 320, 8, 4, 1, None, 0, True),     # BASELINE configs[4] level 0 (768^2, 4 objects; per head)
    (9216, 320, 8, 2, 2, None, 0, True),     # 768^2 with 2 objects: head pairs
    (1024, 640, 8, 2, 2, None, 0, True),     # SD-v1 level 1 (d = 80): Wq + the two mandatory contexts fill the LDS, locals from L2
    (1024, 640, 8, 2, 16, None, 0, True),    # ... 16 prompts: 8 tiles per image, one workgroup per (head, tile pair)
    (1000, 640, 8, 3, 2, 3, 0, True),        # ... ragged N, three objects, forced tile count
    (1024, 640, 8, 1, 3, None, 4, True),     # ... one object, 4-wave workgroups (falls back to 8: the variant is built for 8 waves)
    (2304, 640, 8, 4, 1, None, 0, True),     # BASELINE configs[4] level 1 (768^2, 4 objects)
    (256, 1280, 8, 2, 4, None, 0, True),     # SD-v1 level 2 (d = 160): Wq STREAMED through an LDS ring, contexts 0 / 1 resident, locals from L2
    (256, 1280, 8, 2, 16, 2, 0, True),       # ... 16 prompts, two tiles per workgroup (the ring runs through the attention phase)
    (64, 1280, 8, 2, 8, None, 0, True),      # the middle block: half of the workgroup's waves have no pixels
    (576, 1280, 8, 4, 2, None, 0, True),     # BASELINE configs[4] level 2 (768^2, 4 objects), ragged last tile
    (100, 1280, 8, 1, 3, None, 0, True),     # ragged N (not a multiple of 16), one object
    (144, 1280, 8, 0, 2, None, 0, True),     # 768^2 middle block, no objects
]


@pytest.mark.parametrize("N,C,heads,K,I,tiles,waves,pair", PROJ_SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fwd_proj_matches_oracle(N, C, heads, K, I, tiles, waves, pair, dtype):
    """y -> to_q -> K+2 attentions -> blend in ONE kernel vs the oracle fed with q = round16(y Wq^T) (what the GEMM
    in front of sta_xattn_fwd would have produced), and vs the unfused GPU path on the same inputs."""
    from sta import lib, ops
    dev = "cuda"
    assert ops.proj_supported(C, heads, 77, K)
    g = torch.Generator().manual_seed(N + C + K)
    wq = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dtype)
    cases = [_case(N, C, heads, K, dtype, seed=70 + i) for i in range(I)]          # "q" of a case plays y here
    y = torch.cat([c[0] for c in cases]).to(dev)
    k = torch.cat([c[1] for c in cases]).to(dev)
    v = torch.cat([c[2] for c in cases]).to(dev)
    mb = torch.stack([ops.mask_bits(c[3]) for c in cases]).to(dev)
    coef = torch.stack([c[4] for c in cases]).to(dev)
    scale = (C // heads) ** -0.5
    if tiles is not None:
        lib.set_option(lib.OPT_STAGED_TILES, tiles)
    if waves:
        lib.set_option(lib.OPT_STAGED_WAVES, waves)
    lib.set_option(lib.OPT_PROJ_PAIR, 1 if pair else 2)      # pair: taken wherever the shape allows, whatever the launch size
    out = ops.xattn_forward_proj(y, ops.pack_wq(wq.to(dev), heads), ops.pack_kv_proj(k, v, heads, n_img=I), mb, coef, scale)
    torch.cuda.synchronize()
    lib.set_option(lib.OPT_STAGED_TILES, 0)
    lib.set_option(lib.OPT_STAGED_WAVES, 0)
    lib.set_option(lib.OPT_PROJ_PAIR, 0)
    q_gemm = torch.nn.functional.linear(y, wq.to(dev))
    unfused, _ = ops.xattn_forward(q_gemm, ops.pack_kv(k, v, heads, n_img=I), mb, coef, scale)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    a, b = out.float(), unfused.float()
    # two 16-bit evaluations of the same formula in different summation orders (a q element may round the other way; the
    # head-pair kernel splits the S^T and PV sums 32 + 8 dims / 64 + 16 keys): each side is held to 4 eps against the oracle
    # below, against each other to the sum of the two bounds
    assert ((a - b).abs() <= 8 * eps * (1.0 + b.abs())).all(), (a - b).abs().max().item()
    for i in sorted({0, I - 1}):
        yi, ki, vi, mi, ci = cases[i]
        q16 = (yi.double() @ wq.double().t()).to(dtype)
        ref = orc.fused_xattn(q16.double(), ki.double(), vi.double(), mi, ci.double(), heads, scale)
        err = (a[2 * i:2 * i + 2].cpu().double() - ref).abs()
        assert (err <= 4 * eps * (1.0 + ref.abs())).all(), (i, err.max().item())
    if C == 640 and N % 16 == 0:
        # the locals-from-L2 kernel reads y in query-fragment order too: same values, same arithmetic, bit-identical result
        assert ops.proj_qfrag_supported(C, heads, 77, K, N, I)
        out_f = ops.xattn_forward_proj(ops.to_qfrag(y), ops.pack_wq(wq.to(dev), heads), ops.pack_kv_proj(k, v, heads, n_img=I), mb, coef, scale, qfrag=True)
        assert torch.equal(out_f, out)


PAIR_SHAPES = [
    # N, C, heads, K, images, forced tiles per workgroup, keys
    (256, 320, 8, 2, 1, None, 77),        # two 128-pixel tiles, one per workgroup
    (4096, 320, 8, 2, 4, None, 77),
    (4096, 320, 8, 2, 32, None, 77),      # the default bench launch: 256 pair workgroups x 16 tiles, one round
    (4096, 320, 8, 1, 3, None, 77),       # one object: ragged tile counts per workgroup
    (4096, 320, 8, 0, 2, None, 77),       # no objects
    (1000, 160, 4, 2, 2, 3, 77),          # ragged N, C = 160 (5 k-steps), 2 head pairs, forced tile count
    (1000, 160, 4, 2, 2, 3, 66),          # fewer keys than rows stored: rows 66 .. 76 are zero, the bias masks them
    (1008, 160, 4, 1, 2, 3, 77),          # N % 16 == 0 but not % 128: a partial last tile in query-fragment order, C = 160
    (9216, 320, 8, 2, 2, None, 77),       # 768^2 with 2 objects
]


@pytest.mark.parametrize("N,C,heads,K,I,tiles,M", PAIR_SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fwd_proj_pair_matches_oracle(N, C, heads, K, I, tiles, M, dtype):
    """Head-pair kernel (csrc/sta_xattn_proj3.hip: single-read operand images, 16x16x32 + 16x16x16 MFMAs, software-pipelined
    LDS reads, work items handed out through an LDS counter) vs the oracle fed with q = round16(y Wq^T), image 0 and the last image, and vs the
    unfused GPU path on every image."""
    from sta import lib, ops
    dev = "cuda"
    g = torch.Generator().manual_seed(N + C + K)
    wq = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dtype)
    cases = [_case(N, C, heads, K, dtype, seed=170 + i, M=M) for i in range(I)]
    y = torch.cat([c[0] for c in cases]).to(dev)
    k = torch.cat([c[1] for c in cases]).to(dev)
    v = torch.cat([c[2] for c in cases]).to(dev)
    mb = torch.stack([ops.mask_bits(c[3]) for c in cases]).to(dev)
    coef = torch.stack([c[4] for c in cases]).to(dev)
    scale = (C // heads) ** -0.5
    if tiles is not None:
        lib.set_option(lib.OPT_STAGED_TILES, tiles)
    lib.set_option(lib.OPT_PROJ_PAIR, 1)
    wqf, kvp = ops.pack_wq(wq.to(dev), heads), ops.pack_kv_proj(k, v, heads, n_img=I)
    out_ofrag = None
    out = ops.xattn_forward_proj(y, wqf, kvp, mb, coef, scale)
    # the same launch reading y in QUERY-FRAGMENT order (what sta_add_layernorm_qfrag writes for it): same MFMAs on the same
    # values in the same order => bit-identical; N % 16 != 0 is refused
    assert ops.proj_qfrag_supported(C, heads, M, K, N, I) == (N % 16 == 0)
    if N % 16 == 0:
        out_f = ops.xattn_forward_proj(ops.to_qfrag(y), wqf, kvp, mb, coef, scale, qfrag=True)
        assert torch.equal(out_f, out)
        if ops.proj_ofrag_supported(C, heads, dtype):
            # ... and leaving its output in OUT-FRAGMENT order (for the fused to_out + LayerNorm pass): the same values, permuted.
            # (This equality is what caught the bf16 instantiation of that mode wrong in round 4: a k = 16 MFMA accumulating into a
            # k = 32 MFMA's result one wait state behind it, a pair hipcc 7.2 does not pad — sta/isa_lint.py pads it at build time now,
            # profiles/r05_hazard_table.md.)
            out_ofrag = ops.from_ofrag(ops.xattn_forward_proj(ops.to_qfrag(y), wqf, kvp, mb, coef, scale, qfrag=True, ofrag=True))
            assert torch.equal(out_ofrag, out), (out_ofrag != out).float().mean().item()
        else:
            with pytest.raises(RuntimeError, match="C = 320"):
                ops.xattn_forward_proj(ops.to_qfrag(y), wqf, kvp, mb, coef, scale, qfrag=True, ofrag=True)
    else:
        with pytest.raises(RuntimeError, match="N % 16"):
            ops.xattn_forward_proj(y, wqf, kvp, mb, coef, scale, qfrag=True)
    torch.cuda.synchronize()
    lib.set_option(lib.OPT_STAGED_TILES, 0)
    lib.set_option(lib.OPT_PROJ_PAIR, 0)
    q_gemm = torch.nn.functional.linear(y, wq.to(dev))
    unfused, _ = ops.xattn_forward(q_gemm, ops.pack_kv(k, v, heads, n_img=I), mb, coef, scale)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    a, b = out.float(), unfused.float()
    assert torch.isfinite(a).all()
    # two 16-bit evaluations of the same formula in different summation orders (a q element may round the other way, the
    # S^T and PV sums are split 32 + 8 dims / 64 + 16 keys here): each is held to 4 eps against the oracle below, against
    # each other to the sum of the two bounds
    assert ((a - b).abs() <= 8 * eps * (1.0 + b.abs())).all(), (a - b).abs().max().item()
    for i in sorted({0, I - 1}):
        yi, ki, vi, mi, ci = cases[i]
        q16 = (yi.double() @ wq.double().t()).to(dtype)
        ref = orc.fused_xattn(q16.double(), ki.double(), vi.double(), mi, ci.double(), heads, scale)
        err = (a[2 * i:2 * i + 2].cpu().double() - ref).abs()
        assert (err <= 4 * eps * (1.0 + ref.abs())).all(), (i, err.max().item(), (err / (1.0 + ref.abs())).max().item() / eps)
        if out_ofrag is not None:      # the out-fragment instantiation against the oracle, same bound
            err = (out_ofrag[2 * i:2 * i + 2].float().cpu().double() - ref).abs()
            assert (err <= 4 * eps * (1.0 + ref.abs())).all(), (i, err.max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mode", ["overflow", "underflow", "one_image"])
def test_fwd_proj_pair_extreme_logits(mode, dtype):
    """The head-pair kernel's OPTIMISTIC softmax (P = exp2(S) without the running maximum; csrc/sta_xattn_proj3.hip::attend3) and its
    guard: where a context's denominator leaves its window ([2^-100, 2^100) bf16, [2^-5, 2^15) fp16) the wave repeats that context on the standard path.  overflow: keys x 300
    => |logit| in the hundreds, exp2 overflows in every wave;  underflow: all-positive queries against all-negative keys => every
    P = 0, the denominator is zero;  one_image: only the second image of the launch is extreme (the first must still match the oracle at
    the usual 4 eps).  With logits this large a 2^-9 relative rounding difference in q moves a logit by several tenths, so against the
    fp64 oracle (fed q = round16(y Wq^T), as everywhere) the extreme images are held to: finite everywhere, bit-identical across the three
    layouts, and inside the ordinary 4 eps of the oracle fed the q the kernel rounds (see below)."""
    from sta import lib, ops
    dev, N, C, heads, K, I = "cuda", 1024, 320, 8, 2, 2
    g = torch.Generator().manual_seed(5)
    wq = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dtype)
    cases = [list(_case(N, C, heads, K, dtype, seed=90 + i)) for i in range(I)]
    extreme = [1] if mode == "one_image" else [0, 1]
    if mode == "underflow":
        wq = wq.abs()
    for i in extreme:
        if mode == "underflow":
            cases[i][0] = (cases[i][0].float().abs() + 0.5).to(dtype)         # y > 0, Wq > 0  =>  q > 0
            cases[i][1] = (-(cases[i][1].float().abs() + 0.5) * 8.0).to(dtype)  # k < 0: logits around -500 ... -1500
        else:
            cases[i][1] = (cases[i][1].float() * 300.0).to(dtype)
    y = torch.cat([c[0] for c in cases]).to(dev)
    k = torch.cat([c[1] for c in cases]).to(dev)
    v = torch.cat([c[2] for c in cases]).to(dev)
    mb = torch.stack([ops.mask_bits(c[3]) for c in cases]).to(dev)
    coef = torch.stack([c[4] for c in cases]).to(dev)
    scale = (C // heads) ** -0.5
    lib.set_option(lib.OPT_PROJ_PAIR, 1)
    try:
        wqf, kvp = ops.pack_wq(wq.to(dev), heads), ops.pack_kv_proj(k, v, heads, n_img=I)
        out = ops.xattn_forward_proj(y, wqf, kvp, mb, coef, scale)
        out_q = ops.xattn_forward_proj(ops.to_qfrag(y), wqf, kvp, mb, coef, scale, qfrag=True)
        out_o = ops.from_ofrag(ops.xattn_forward_proj(ops.to_qfrag(y), wqf, kvp, mb, coef, scale, qfrag=True, ofrag=True))
    finally:
        lib.set_option(lib.OPT_PROJ_PAIR, 0)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert torch.equal(out_q, out) and torch.equal(out_o, out)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    sl2e = scale * 1.4426950408889634
    for i in range(I):
        yi, ki, vi, mi, ci = cases[i]
        q64 = yi.double() @ wq.double().t()
        ref = orc.fused_xattn(q64.to(dtype).double(), ki.double(), vi.double(), mi, ci.double(), heads, scale)
        frac = lambda r: ((out[2 * i:2 * i + 2].float().cpu().double() - r).abs() <= 4 * eps * (1.0 + r.abs())).double().mean().item()
        if i in extreme:
            # the kernel folds scale * log2(e) into its Wq fragments (W' = round16(W scale log2 e): scores in log2 units) and rounds
            # q = y W'^T to bf16, the oracle's usual input is round16(y W^T): at logits in the hundreds that 2^-9 relative difference moves a
            # logit by several tenths and flips near-tie pixels (the reference's own bf16 logits carry the same relative error). Held to 4 eps against the oracle fed the q the kernel uses; the usual input: printed
            q_k = (yi.double() @ (wq.double() * sl2e).to(dtype).double().t()).to(dtype).double() / sl2e      # W' = round16(W scale log2 e), q = round16(y W'^T)
            ref_k = orc.fused_xattn(q_k, ki.double(), vi.double(), mi, ci.double(), heads, scale)
            print("%s, image %d: inside 4 eps of the oracle fed q rounded in log2 units %.4f, fed round16(q) %.4f" % (mode, i, frac(ref_k), frac(ref)))
            assert frac(ref_k) > 0.995, (i, frac(ref_k), frac(ref))
        else:
            assert frac(ref) == 1.0, (i, frac(ref))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fwd_proj_pair_counts_fallbacks_and_sits_out(dtype):
    """sta_xattn_fwd_proj_ex's statistics words: friendly logits count evaluations and no fall-back; with a key that leads by +16 nats
    ("BOS") every evaluation of the fp16 window falls back, the launch after it sits the optimistic softmax out (state word 64 -> 63 ...),
    a launch without statistics words gives the optimistic launch's bits and the sitting-out launch the same values to output rounding."""
    from sta import lib, ops
    dev, N, C, heads, K, I = "cuda", 4096, 320, 8, 2, 2
    g = torch.Generator().manual_seed(6)
    wq = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dtype).to(dev)
    cases = [list(_case(N, C, heads, K, dtype, seed=120 + i)) for i in range(I)]
    y = torch.cat([c[0] for c in cases]).to(dev)
    v = torch.cat([c[2] for c in cases]).to(dev)
    mb = torch.stack([ops.mask_bits(c[3]) for c in cases]).to(dev)
    coef = torch.stack([c[4] for c in cases]).to(dev)
    scale = (C // heads) ** -0.5
    wqf = ops.pack_wq(wq, heads)
    q = (y.float() @ wq.float().t())                                       # [2I, N, C]
    lib.set_option(lib.OPT_PROJ_PAIR, 1)
    try:
        for hostile in (False, True):
            k = torch.cat([c[1] for c in cases]).float()
            if hostile:
                # key 0 of every context follows the mean query of its head, scaled so that its logit leads by ~ +16 nats (fp16's window
                # ends near +10, bf16's near +65: bf16 keeps the optimistic path)
                qm = q.view(2 * I, N, heads, C // heads).mean((0, 1))          # [heads, d]
                qm = qm / qm.norm(dim=-1, keepdim=True)
                lead = 16.0 / scale / (q.view(2 * I, N, heads, C // heads).float() * qm).sum(-1).abs().mean().item()
                k[:, 0] = (qm * lead).reshape(C).cpu()
            kvp = ops.pack_kv_proj(k.to(dtype).to(dev), v, heads, n_img=I)
            stats = torch.zeros(lib.P3_STATS_WORDS, dtype=torch.int32, device=dev)
            yq = ops.to_qfrag(y)
            first = ops.xattn_forward_proj(yq, wqf, kvp, mb, coef, scale, qfrag=True, ofrag=True, stats=stats)
            s1 = stats.cpu().tolist()
            second = ops.xattn_forward_proj(yq, wqf, kvp, mb, coef, scale, qfrag=True, ofrag=True, stats=stats)
            s2 = stats.cpu().tolist()
            plain = ops.xattn_forward_proj(yq, wqf, kvp, mb, coef, scale, qfrag=True, ofrag=True, stats=None)
            torch.cuda.synchronize()
            assert s1[1] == 0 and s1[2] == 0 and s1[3] == 0 and s1[6] == 1 and s1[4] > 0        # counts folded into the totals by the launch's last sampled workgroup
            # a launch with statistics words takes the same path per context as one without; a launch that sits out takes the standard softmax
            # for EVERY context (another rounding of P where the optimistic path had passed its check): the same values to output rounding
            eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
            assert torch.equal(first, plain) and torch.isfinite(first).all()
            assert ((second.float() - first.float()).abs() <= 2 * eps * (1.0 + first.float().abs())).all()
            rate = s1[5] / s1[4]
            if hostile and dtype == torch.float16:
                assert rate > 0.5 and s1[0] == 64 and s2[0] == 63 and s2[7] == 1 and s2[4] == s1[4], (s1, s2)    # the second launch evaluated nothing optimistically
            else:                # (bf16's window ends near +65 nats: of the hostile case only the pixels that project > 4x the mean onto the key leave it)
                assert (s1[5] == 0 or (hostile and rate < 0.05)) and s1[0] == 0 and s2[4] == 2 * s1[4] and s2[7] == 0, (s1, s2)
    finally:
        lib.set_option(lib.OPT_PROJ_PAIR, 0)


def test_fwd_proj_rejects_what_it_cannot_hold():
    """Level 2 of SD-v1 (C = 1280: 400 KiB of Wq per head) does not fit one CU's LDS, and level 1 (C = 640: 100 KiB of Wq + 4
    contexts of 30 KiB) fits only with the local contexts left in L2 (STA_OPT_PROJ_LL2 = 2 refuses that variant): the C-ABI
    says so instead of launching, and the block then takes the GEMM + sta_xattn_fwd."""
    from sta import lib, ops
    assert ops.proj_supported(640, 8, 77, 2) and ops.proj_supported(640, 8, 77, 0)
    # C = 1280 (d = 160) is held by the streamed-Wq kernel (built, measured slower than the GEMM + sta_xattn_fwd, not used by the model:
    # sta.ops.PROJ_WQS_IN_MODEL); d = 144 has no kernel
    assert ops.proj_supported(1280, 8, 77, 2) and ops.proj_streams_wq(1280, 8) and not ops.PROJ_WQS_IN_MODEL
    assert not ops.proj_supported(1152, 8, 77, 2)
    assert not ops.proj_supported(320, 8, 77, 5)
    L = lib.load()
    y = torch.zeros(2, 64, 640, device="cuda", dtype=torch.float16)
    lib.set_option(lib.OPT_PROJ_LL2, 2)
    try:
        assert not ops.proj_supported(640, 8, 77, 2)
        rc = L.sta_xattn_fwd_proj(y.data_ptr(), y.data_ptr(), y.data_ptr(), y.data_ptr(), y.data_ptr(), y.data_ptr(), 1, 64, 640, 8, 77, 2,
                                  1.0, lib.STA_F16, 0)
        err = lib.last_error()
    finally:
        lib.set_option(lib.OPT_PROJ_LL2, 0)
    assert rc == -2 and "LDS" in err


def test_toolchain_self_check_passes_and_is_wired():
    """sta.ops.toolchain_self_check: the level-0 head-pair kernel in its three layouts against the one-head-per-workgroup kernel, both
    16-bit types — what a library built by a hipcc release other than the validated one runs before its first projection-fused launch
    (sta.lib.toolchain_validated reads the release from the library: sta_built_with). The shipped library must pass it."""
    from sta import lib, ops
    assert lib.built_with().startswith("HIP version")
    assert ops.toolchain_self_check(force=True) is True
    assert ops._TOOLCHAIN_CHECKED is True
