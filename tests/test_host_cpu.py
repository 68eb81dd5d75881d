"""Host logic of the product (blocks, UNet, sampler, schedule, C-ABI loading) on CPU.

The two HIP entry points are swapped for the oracle's fused form (tests/cpu_backend.py) so that the
Python around them can be checked against the reference's golden vectors without a GPU. The real
kernels are checked in the `-m gpu` tests.
"""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import golden_inputs as gi
from sta.synth import seeded_fill_
from tests.cpu_backend import oracle_ops

G = gi.GOLDEN
REPO = os.path.dirname(G.rstrip("/")).rsplit("/tests", 1)[0]


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_library_loads_and_exports_every_declared_symbol():
    from sta import lib
    header = open(os.path.join(lib.INCLUDE, "sta_xattn.h")).read() + open(os.path.join(lib.INCLUDE, "sta_unet.h")).read()
    declared = set(re.findall(r"\b(sta_[a-z_0-9]+)\s*\(", header))
    assert declared == set(lib.SYMBOLS), (declared, set(lib.SYMBOLS))
    L = lib.load()                       # raises if the .so is missing or a symbol is absent
    assert L.sta_version() == 0x000500
    assert L.sta_last_error() == b""
    # host-only entry points (no GPU work)
    assert L.sta_xattn_packed_kv_bytes(4, 8, 40) == 4 * 8 * 2 * (5 * 2 + 3 * 3) * 1024
    assert L.sta_xattn_packed_kv_bytes(4, 8, 168) == 0 and L.sta_xattn_packed_kv_bytes(4, 8, 20) == 0
    assert L.sta_selfattn_fwd(0, 0, 0, 0, 2, 64, 64, 8, 64, 64, 64, 4096, 1.0, 0, 0) == -1 and b"null" in L.sta_last_error()
    assert L.sta_selfattn_fwd_lse(8, 8, 8, 8, 0, 2, 64, 64, 8, 64, 64, 64, 4096, 1.0, 0, 0) == -1 and b"null" in L.sta_last_error()
    assert L.sta_selfattn_bwd(*([0] * 13), 2, 64, 64, 8, 64, 64, 1.0, 0, 0) == -1 and b"null" in L.sta_last_error()
    # the backward streams whole 64-row blocks: N % 64 != 0 is refused before anything is launched
    assert L.sta_selfattn_bwd(*([8] * 13), 2, 72, 64, 8, 64, 64, 1.0, 0, 0) == -2 and b"N % 64" in L.sta_last_error()
    assert L.sta_groupnorm_silu(0, 0, 0, 0, 0, 2, 320, 4096, 32, 1e-5, 1, 0, 0) == -1 and b"null" in L.sta_last_error()
    assert L.sta_geglu(0, 0, 4, 64, 0, 0) == -1 and L.sta_add_layernorm(0, 0, 0, 0, 0, 0, 0, 4, 64, 1e-5, 0, 0) == -1
    assert L.sta_add_bias_nchw(0, 0, 0, 0, 2, 4, 64, 0, 0) == -1 and L.sta_add_bias_rows(0, 0, 0, 0, 8, 64, 0, 0) == -1
    assert L.sta_groupnorm_nhwc_workspace_bytes(16, 4096, 32) == 16 * 32 * 2 * 32 * 4
    assert L.sta_groupnorm_silu_nhwc(0, 0, 0, 0, 0, 0, 2, 320, 4096, 32, 1e-5, 1, 0, 0) == -1
    assert L.sta_groupnorm_silu_nhwc_bwd(0, 0, 0, 0, 0, 0, 0, 0, 2, 320, 4096, 32, 1e-5, 1, 0, 0) == -1 and b"null" in L.sta_last_error()
    assert L.sta_geglu_bwd(0, 0, 0, 4, 64, 0, 0) == -1 and L.sta_layernorm_bwd(0, 0, 0, 0, 0, 4, 64, 1e-5, 0, 0) == -1
    assert L.sta_xattn_bwd_workspace_bytes(1, 4096, 8, 2) >= 2 * 256 * 8 * 4
    assert L.sta_xattn_bwd_workspace_bytes(3, 4096, 8, 2) == 3 * L.sta_xattn_bwd_workspace_bytes(1, 4096, 8, 2)


def test_missing_gpu_fails_loudly():
    from sta import ops
    q = torch.zeros(2, 16, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.pack_kv(torch.zeros(2, 77, 64, dtype=torch.bfloat16), torch.zeros(2, 77, 64, dtype=torch.bfloat16), 8)
    packed = ops.PackedKV(torch.zeros(1, dtype=torch.uint8), 2, 8, 77, 64, torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.xattn_blend(q, None, packed, None, 1.0)
    # the differentiable self-attention refuses CPU tensors and shapes its backward cannot take, at forward time
    with pytest.raises(ValueError, match="SelfAttentionQKV needs a CUDA"):
        ops.SelfAttentionQKV.apply(torch.zeros(2, 64, 3 * 64, dtype=torch.float16), 8, 0.35)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.self_attention(q, q, torch.zeros(2, 64, 16, dtype=torch.bfloat16), 8, 0.35)


def test_product_disc_masks_bit_exact():
    from sta import ops
    g = _load("masks.npz")
    centres = [tuple(c) for c in g["centres"]]
    for dim in g["dims"]:
        ref = np.unpackbits(g["mask_%d" % dim], axis=1)[:, : dim * dim]
        got = ops.disc_masks(centres, int(dim)).numpy()
        assert got.dtype == np.uint8 and (got == ref).all(), dim
        bits = ops.disc_mask_bits(centres, int(dim)).numpy()          # what the kernels read: bit i = disc i
        assert bits.dtype == np.uint8 and all((((bits >> i) & 1) == ref[i]).all() for i in range(len(centres)))


@pytest.mark.parametrize("name", ["d40", "d80", "d160", "d8k4", "k0"])
@pytest.mark.parametrize("use_ckpt", [False, True])
def test_block_forward_and_dcoef(name, use_ckpt):
    from ldm.modules.attention import BasicTransformerBlock
    from sta import prompt_state
    g = _load("block_%s.npz" % name)
    dim, C, heads, K, seed = (int(g[k]) for k in ("dim", "C", "heads", "K", "seed"))
    x, context, local_ctx = gi.block_inputs(dim, C, K, seed, gi.load_uncond())
    blk = BasicTransformerBlock(C, heads, C // heads, context_dim=768, checkpoint=use_ckpt)
    assert abs(seeded_fill_(blk, seed) - float(g["checksum"])) <= 1e-6 * float(g["checksum"])
    for p in blk.parameters():
        p.requires_grad_(False)
    coef = torch.from_numpy(g["coef"]).requires_grad_(K > 0)
    centres = [list(c) for c in g["centres"]]
    with oracle_ops():
        prompt_state.begin_prompt(local_ctx, first_timestep=981)
        out = blk(x.clone(), context=context, time=torch.tensor(981), coef=coef, bboxs_curr=centres)
        np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=0, atol=5e-4)
        if K:
            (0.5 * (out * out).sum()).backward()
            np.testing.assert_allclose(coef.grad.numpy(), g["dcoef"], rtol=1e-3)


def test_block_state_follows_prompt_version():
    """A new prompt (begin_prompt) must re-pack K/V; the same prompt must not."""
    from ldm.modules.attention import BasicTransformerBlock
    from sta import ops, prompt_state
    blk = BasicTransformerBlock(64, 8, 8, context_dim=768, checkpoint=False)
    x, ctx = torch.randn(2, 64, 64), torch.randn(2, 77, 768)
    calls = []
    with oracle_ops():
        real = ops.pack_kv
        ops.pack_kv = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        for prompt in range(2):
            prompt_state.begin_prompt([torch.randn(1, 77, 768)], first_timestep=981)
            for step in range(3):
                blk(x, context=ctx, time=torch.tensor(981 - 20 * step), coef=torch.ones(1), bboxs_curr=[[0.5, 0.5]])
        assert len(calls) == 2
        with pytest.raises(ValueError, match=r"announced \[1\] local prompts"):
            blk(x, context=ctx, time=torch.tensor(1), coef=torch.ones(2), bboxs_curr=[[0.5, 0.5], [0.2, 0.2]])
        with pytest.raises(ValueError, match="batch 2 per image"):
            blk(x[:1], context=ctx, time=torch.tensor(1), coef=torch.ones(1), bboxs_curr=[[0.5, 0.5]])


def test_block_batch_of_images_equals_one_by_one():
    """n_img images per call (rows interleaved [uncond_i, cond_i]) == the same images one at a time."""
    from ldm.modules.attention import BasicTransformerBlock
    from sta import prompt_state
    blk = BasicTransformerBlock(64, 8, 8, context_dim=768, checkpoint=False)
    seeded_fill_(blk, 4)
    g = torch.Generator().manual_seed(0)
    I, K, N = 3, 2, 64
    x, ctx = torch.randn(2 * I, N, 64, generator=g), torch.randn(2 * I, 77, 768, generator=g)
    local = [[torch.randn(1, 77, 768, generator=g) for _ in range(K)] for _ in range(I)]
    boxes = [[[0.3 + 0.1 * i, 0.4], [0.7, 0.6 - 0.1 * i]] for i in range(I)]
    coef = torch.rand(I, K, generator=g) + 1
    with oracle_ops(), torch.no_grad():
        prompt_state.begin_prompt(local, first_timestep=981)
        batched = blk(x, context=ctx, time=torch.tensor(981), coef=coef, bboxs_curr=boxes)
        for i in range(I):
            prompt_state.begin_prompt(local[i], first_timestep=981)
            one = blk(x[2 * i:2 * i + 2], context=ctx[2 * i:2 * i + 2], time=torch.tensor(981), coef=coef[i], bboxs_curr=boxes[i])
            assert torch.allclose(batched[2 * i:2 * i + 2], one, atol=1e-5)
        with pytest.raises(ValueError, match="one box list per image"):
            blk(x, context=ctx, time=torch.tensor(981), coef=coef, bboxs_curr=boxes[:2])


def _golden_unet():
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    meta = json.load(open(os.path.join(G, "unet_state_dict_keys.json")))
    unet = UNetModel(**meta["cfg"]).eval()
    return unet, meta


def test_unet_state_dict_contract():
    unet, meta = _golden_unet()
    ours = {k: list(v.shape) for k, v in unet.state_dict().items()}
    assert ours == meta["shapes"]          # same keys, same shapes as the reference's UNetModel


def test_unet_eps_matches_reference():
    from sta import prompt_state
    g = _load("unet_eps.npz")
    unet, _ = _golden_unet()
    assert abs(seeded_fill_(unet, 21) - float(g["checksum"])) <= 1e-6 * float(g["checksum"])
    c, local_ctx, _ = gi.unet_inputs(2, int(g["input_seed"]))
    uncond = gi.load_uncond()
    with oracle_ops(), torch.no_grad():
        prompt_state.begin_prompt(local_ctx, first_timestep=981)
        eps = unet(torch.from_numpy(g["x_in"]), 0, torch.from_numpy(g["t"]), context=torch.cat([uncond, c]),
                   coef=torch.from_numpy(g["coef"]), bboxs_curr=[list(cc) for cc in g["centres"]])
    np.testing.assert_allclose(eps.numpy(), g["eps"], rtol=0, atol=2e-4)


def test_sampler_schedule_tables():
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from ldm.models.diffusion.plms import PLMSSampler
    g = _load("schedule.npz")
    unet, _ = _golden_unet()
    model = LatentDiffusion(unet_config=unet)
    np.testing.assert_allclose(model.alphas_cumprod.numpy(), g["alphas_cumprod"], rtol=1e-6)
    for S in (50, 10):
        s = PLMSSampler(model, opt_epochs=0)
        s.make_schedule(S, verbose=False)
        assert (s.ddim_timesteps == g["t_%d" % S]).all()
        np.testing.assert_allclose(s.ddim_alphas, g["a_%d" % S], rtol=1e-6)
        np.testing.assert_allclose(s.ddim_alphas_prev, g["ap_%d" % S], rtol=1e-6)
        np.testing.assert_allclose(s.ddim_sqrt_one_minus_alphas, g["s1m_%d" % S], rtol=1e-6)
    with pytest.raises(ValueError, match="ddim_eta must be 0"):
        PLMSSampler(model).make_schedule(50, ddim_eta=0.5)


def test_plms_trajectory_matches_reference():
    """50 PLMS steps (51 UNet calls) with CFG 7.5 and a different weight column per step."""
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from ldm.models.diffusion.plms import PLMSSampler
    from sta import prompt_state
    g = _load("plms_traj.npz")
    unet, _ = _golden_unet()
    seeded_fill_(unet, 21)
    model = LatentDiffusion(unet_config=unet)
    c, local_ctx, x_T = gi.unet_inputs(2, int(g["input_seed"]))
    np.testing.assert_array_equal(x_T.numpy(), g["x_T"])
    uncond = gi.load_uncond()
    centres = [list(cc) for cc in g["centres"]]
    W = torch.from_numpy(g["W"])
    S, scale = int(g["S"]), float(g["scale"])
    sampler = PLMSSampler(model, opt_epochs=0, use_graph=False, save_images=False)
    sampler.make_schedule(S, verbose=False)
    time_range = np.flip(sampler.ddim_timesteps)
    keep = list(g["keep"])
    with oracle_ops(), torch.no_grad():
        prompt_state.begin_prompt(local_ctx, first_timestep=int(time_range[0]))
        eps_fn = sampler._make_eps_fn(c, uncond, scale, centres, 0, False, x_T)
        img, old_eps = x_T.clone(), []
        for i, step in enumerate(time_range):
            ts = torch.full((1,), int(step), dtype=torch.long)
            tn = torch.full((1,), int(time_range[min(i + 1, S - 1)]), dtype=torch.long)
            img, _, e_t = sampler._plms_update(eps_fn, img, ts, tn, S - i - 1, old_eps, W[:, i])
            old_eps.append(e_t)
            if len(old_eps) >= 4:
                old_eps.pop(0)
            if i == 0:
                np.testing.assert_allclose(e_t.numpy(), g["e0"], rtol=0, atol=1e-3)
            if i in keep:
                ref = g["xs"][keep.index(i)]
                err = np.abs(img.numpy() - ref).max() / max(1.0, np.abs(ref).max())
                assert err < 2e-3, (i, err)
    err = np.abs(img.numpy() - g["x0"]).max() / np.abs(g["x0"]).max()
    assert err < 2e-3, err


def test_sampler_fixed_weights_path_and_result():
    """sample(...) with opt_epochs=0: reference keyword surface, W = 5/K, result kept for callers."""
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from ldm.models.diffusion.plms import PLMSSampler
    unet, _ = _golden_unet()
    seeded_fill_(unet, 21)
    model = LatentDiffusion(unet_config=unet)
    c, local_ctx, x_T = gi.unet_inputs(2, 5)
    sampler = PLMSSampler(model, opt_epochs=0, use_graph=False, save_images=False)
    with oracle_ops():
        out = sampler.sample(S=10, conditioning=c, batch_size=1, shape=[4, 16, 16], verbose=False,
                             unconditional_guidance_scale=7.5, unconditional_conditioning=gi.load_uncond(), eta=0.0,
                             x_T=x_T[:, :, :16, :16], text_index=0, curr_text="x", bboxs_curr=[[0.3, 0.4], [0.7, 0.6]],
                             seed=1, prompt_idx=0, object_names=["a", "b"], local_conditionings=local_ctx)
    assert out is None
    r = sampler.last_result
    assert r["x0"].shape == (1, 4, 16, 16) and torch.isfinite(r["x0"]).all()
    assert r["W"].shape == (2, 10) and torch.allclose(r["W"], torch.full((2, 10), 2.5))
    with pytest.raises(AssertionError):
        sampler.sample(S=10, conditioning=c, batch_size=1, shape=[4, 16, 16], verbose=False, unconditional_guidance_scale=7.5,
                       unconditional_conditioning=gi.load_uncond(), bboxs_curr=[[0.3, 0.4]], object_names=[], seed=1)


def test_sample_batch_equals_prompt_by_prompt():
    """sample_batch over I prompts == sample(...) on each prompt alone (independent prompts, CFG pairs adjacent)."""
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from ldm.models.diffusion.plms import PLMSSampler
    unet, _ = _golden_unet()
    seeded_fill_(unet, 21)
    model = LatentDiffusion(unet_config=unet)
    I, K = 2, 2
    cs, locs, xts, boxes = [], [], [], []
    for i in range(I):
        c, local_ctx, x_T = gi.unet_inputs(K, 50 + i)
        cs.append(c), locs.append(local_ctx), xts.append(x_T[:, :, :16, :16])
        boxes.append([[0.3 + 0.2 * i, 0.4], [0.7, 0.6 - 0.2 * i]])
    uc = gi.load_uncond()
    sampler = PLMSSampler(model, opt_epochs=0, use_graph=False, save_images=False)
    with oracle_ops():
        sampler.sample_batch(S=5, shape=[4, 16, 16], conditionings=cs, unconditional_conditionings=uc, bboxs=boxes,
                             object_names=[["a", "b"]] * I, local_conditionings=locs, x_T=torch.cat(xts), seed=1)
        batch = sampler.last_result["x0"].clone()
        assert batch.shape == (I, 4, 16, 16) and sampler.last_result["W"].shape == (I, K, 5)
        for i in range(I):
            sampler.sample(S=5, conditioning=cs[i], batch_size=1, shape=[4, 16, 16], verbose=False, unconditional_guidance_scale=7.5,
                           unconditional_conditioning=uc, eta=0.0, x_T=xts[i], text_index=0, curr_text="x", bboxs_curr=boxes[i],
                           seed=1, prompt_idx=i, object_names=["a", "b"], local_conditionings=locs[i])
            one = sampler.last_result["x0"]
            assert torch.allclose(batch[i:i + 1], one, atol=1e-4 * float(one.abs().max())), (i, (batch[i:i + 1] - one).abs().max())


def test_weight_optimisation_epochs_move_W():
    """opt_epochs=2 with a differentiable stand-in loss: W gets one Adam step (lr 5e-3) from the first
    epoch; every column receives gradient only from its own step's UNet call(s)."""
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from ldm.models.diffusion.plms import PLMSSampler
    from ldm.models.autoencoder import AutoencoderKL

    class Loss(torch.nn.Module):
        def forward_2(self, image, text):
            return image.mean().reshape(1)

        def forward_3(self, image, text):
            return (image ** 2).mean().reshape(1)

    unet, _ = _golden_unet()
    seeded_fill_(unet, 21)
    for p in unet.parameters():
        p.requires_grad_(False)
    vae = AutoencoderKL(ddconfig=dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32,
                                      ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[], dropout=0.0))
    seeded_fill_(vae, 3)
    for p in vae.parameters():
        p.requires_grad_(False)
    model = LatentDiffusion(unet_config=unet, first_stage_config=vae)
    c, local_ctx, x_T = gi.unet_inputs(2, 6)
    sampler = PLMSSampler(model, loss_model=Loss(), opt_epochs=2, use_graph=False, save_images=False)
    with oracle_ops():
        sampler.sample(S=5, conditioning=c, batch_size=1, shape=[4, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
                       unconditional_conditioning=gi.load_uncond(), x_T=x_T[:, :, :8, :8], text_index=0, curr_text="x",
                       bboxs_curr=[[0.3, 0.4], [0.7, 0.6]], seed=1, prompt_idx=0, object_names=["The cat", "dog"],
                       local_conditionings=local_ctx)
    r = sampler.last_result
    assert len(r["losses"]) == 1 and r["image"].shape == (1, 3, 16, 16)
    step = (r["W"] - 2.5).abs()
    assert torch.allclose(step, torch.full_like(step, 0.005), atol=1e-4)   # first Adam step has size lr everywhere


def test_loss_front_end_matches_reference():
    """G8: the product's DCLIPLoss front-ends and `_fidelity_loss` (crop rule, object-name normalisation, weight 5) vs
    the REFERENCE's DCLIPLoss.forward_2 / forward_3 (plms.py:21-45) run around the same frozen stand-in CLIP, on the
    crops of plms.py:254-270: the 224^2 images handed to CLIP, each loss term and the total."""
    from ldm.models.diffusion.plms import DCLIPLoss, PLMSSampler, object_crop_box
    from oracle.gen_golden import LOSS_CASES
    from sta.synth import SyntheticCLIP
    g = _load("loss_frontend.npz")
    model = SyntheticCLIP()
    fed = []
    enc = model.encode_image
    model.encode_image = lambda img: (fed.append(img.detach().clone()), enc(img))[1]
    lm = DCLIPLoss(model)
    sampler = object.__new__(PLMSSampler)
    sampler.clip_loss_model, sampler.local_loss_weight = lm, 5.0
    assert int(g["n_cases"]) == len(LOSS_CASES)
    for n, (seed, text, objs) in enumerate(LOSS_CASES):
        img = gi.loss_image(seed)
        assert abs(float(img.double().sum()) - float(g["case%d_img_sum" % n])) < 1e-3        # same seeded image
        del fed[:]
        with torch.no_grad():
            total = sampler._fidelity_loss(img, text, [c for _, c in objs], [nm for nm, _ in objs])
        boxes = [object_crop_box(c, 512, 512) for _, c in objs]
        assert np.array_equal(np.asarray(boxes, dtype=np.int64).reshape(-1, 4), g["case%d_boxes" % n])
        assert len(fed) == 1 + len(objs)
        got = np.stack([f[0, :, ::7, ::7].numpy() for f in fed])
        assert np.abs(got - g["case%d_fed" % n]).max() < 1e-6
        assert np.allclose([float(f.double().sum()) for f in fed], g["case%d_fed_sum" % n], rtol=1e-6)
        assert abs(float(total) - float(g["case%d_total" % n])) < 1e-5, (float(total), float(g["case%d_total" % n]))
        with torch.no_grad():
            assert abs(float(lm.forward_2(img, text)) - float(g["case%d_loss2" % n])) < 1e-6
            for (nm, c), b, ref in zip(objs, boxes, g["case%d_loss3" % n]):
                l3 = lm.forward_3(img[:, b[0]:b[1], b[2]:b[3]], "A photo of " + nm.lower().replace("the ", ""))
                assert abs(float(l3) - float(ref)) < 1e-6


def test_loss_model_is_checked_before_sampling():
    """opt_epochs > 0 without a loss model fails BEFORE the first trajectory; DCLIPLoss tokenises through
    the callable it is given (the reference's clip.tokenize, plms.py:31,38)."""
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from ldm.models.diffusion.plms import DCLIPLoss, PLMSSampler, load_clip_model
    from sta.synth import SyntheticCLIP
    unet, _ = _golden_unet()
    model = LatentDiffusion(unet_config=unet)
    calls = []
    model.apply_model_extra = lambda *a, **k: calls.append(1)
    c, local_ctx, x_T = gi.unet_inputs(2, 6)
    sampler = PLMSSampler(model, opt_epochs=3, use_graph=False, save_images=False)
    with pytest.raises(RuntimeError, match="loss_model"):
        sampler.sample(S=5, conditioning=c, batch_size=1, shape=[4, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
                       unconditional_conditioning=gi.load_uncond(), x_T=x_T[:, :, :8, :8], text_index=0, curr_text="x",
                       bboxs_curr=[[0.3, 0.4], [0.7, 0.6]], seed=1, prompt_idx=0, object_names=["a", "b"],
                       local_conditionings=local_ctx)
    assert not calls
    seen = []
    clip = SyntheticCLIP()
    lm = DCLIPLoss(clip, tokenize=lambda texts: (seen.append(list(texts)), torch.zeros(1, 77, dtype=torch.long))[1])
    clip.encode_text = lambda tok: (seen.append(tuple(tok.shape)), torch.ones(1, 512))[1]
    lm.forward_2(torch.rand(3, 512, 512), "a cat")
    assert seen == [["a cat"], (1, 77)]
    with pytest.raises(RuntimeError, match="CLIP"):
        load_clip_model(None, "cpu")                      # the OpenAI package is absent here: a clear error, not a late crash
    m, tok = load_clip_model("sta.synth:synthetic_clip", "cpu")      # module:callable form
    assert hasattr(m, "encode_image")


def test_config1_trajectory_matches_reference():
    """BASELINE configs[0] (G5b): one prompt, 64x64 latent, 10 PLMS steps, 1 object, through the PUBLIC sample() entry
    (fixed weights W = 5/K, first timestep 901 announced to the blocks instead of the reference's `time == 981` test)
    vs the reference's fp32 CPU trajectory."""
    from ldm.models.diffusion.ddpm import LatentDiffusion
    from ldm.models.diffusion.plms import PLMSSampler
    g = _load("plms_config1.npz")
    unet, _ = _golden_unet()
    seeded_fill_(unet, 21)
    model = LatentDiffusion(unet_config=unet)
    K, S, lat = int(g["K"]), int(g["S"]), int(g["lat"])
    c, local_ctx, x_T = gi.unet_inputs(K, int(g["input_seed"]), lat)
    assert abs(float(x_T.double().abs().sum()) - float(g["x_T_sum"])) < 1e-6 * float(g["x_T_sum"])
    sampler = PLMSSampler(model, opt_epochs=0, use_graph=False, save_images=False)
    with oracle_ops():
        sampler.sample(S=S, conditioning=c, batch_size=1, shape=[4, lat, lat], verbose=False, unconditional_guidance_scale=float(g["scale"]),
                       unconditional_conditioning=gi.load_uncond(), eta=0.0, x_T=x_T, text_index=0, curr_text="a prompt",
                       bboxs_curr=[list(cc) for cc in g["centres"]], seed=1, prompt_idx=0, object_names=["thing"],
                       local_conditionings=local_ctx)
    x0 = sampler.last_result["x0"].numpy()
    err = np.abs(x0 - g["x0"]).max() / np.abs(g["x0"]).max()
    assert err < 2e-3, err


def test_calls_to_keep_is_sized_to_free_memory(monkeypatch):
    """Per-call recomputation keeps the activations of as many trailing UNet calls as HBM holds (PLMSSampler._calls_to_keep):
    none before the per-call size is known (then one, to measure it), afterwards (free - 4 calls - 24 GiB) / size, clamped."""
    from ldm.models.diffusion.plms import PLMSSampler
    s = PLMSSampler.__new__(PLMSSampler)
    s.keep_calls, s._call_key = None, None
    gib = 1 << 30
    assert s._calls_to_keep(51, 16) == 0                                   # no GPU: nothing to size against
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda: (200 * gib, 288 * gib))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda: 30 * gib)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda: 10 * gib)
    assert s._calls_to_keep(51, 16) == 1                                   # first tracked epoch: one call, to measure its size
    s._call_bytes_per_image = gib                                          # 16 GiB per call at 16 prompts
    s._call_key = ((4, 64, 64), 2)
    assert s._calls_to_keep(51, 16) == 1                                   # measured for ANOTHER (latent shape, K): measure again
    s._call_bytes_key = s._call_key
    assert s._calls_to_keep(51, 16) == (220 - 4 * 16 - 24) // 16           # free = 200 + cached 20 GiB
    assert s._calls_to_keep(51, 1) == 51                                   # one prompt: everything fits -> no recomputation at all
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda: (40 * gib, 288 * gib))
    assert s._calls_to_keep(51, 16) == 0
    s.keep_calls = 5
    assert s._calls_to_keep(51, 16) == 5 and s._calls_to_keep(3, 16) == 3
    s.keep_calls = 0
    assert s._calls_to_keep(51, 16) == 0


def test_fp16_loss_scale_follows_the_loss_value():
    """PLMSSampler._loss_scale: fp16 models scale the loss to [2^15, 2^16) with a power of two, other types do not scale; an explicit
    loss_scale wins."""
    import types
    from ldm.models.diffusion.plms import PLMSSampler
    s = PLMSSampler.__new__(PLMSSampler)
    s.loss_scale = None
    s.model = types.SimpleNamespace(model=torch.nn.Linear(2, 2).half())
    for loss in (11.18, 174.9, 1.7e-4, 0.7):
        sc = s._loss_scale(loss)
        assert sc == 2.0 ** round(np.log2(sc)) and 2.0 ** 15 <= loss * sc < 2.0 ** 16, (loss, sc)
    assert s._loss_scale(11.18) == 4096.0
    assert s._loss_scale(0.0) == 1.0 and s._loss_scale(float("inf")) == 1.0 and s._loss_scale(1e9) == 1.0
    s.model = types.SimpleNamespace(model=torch.nn.Linear(2, 2).bfloat16())
    assert s._loss_scale(11.18) == 1.0
    s.loss_scale = 128.0
    assert s._loss_scale(11.18) == 128.0


def test_trunk_kernel_host_entry_points():
    """Host-only logic of include/sta_unet.h's convolution / row GEMM entry points (geometry rules, sizes, argument checks): no GPU work."""
    from sta import fused, lib
    L = lib.load()
    # 3x3 convolution: 8 x 32 tiles, 16 x 16 tiles, whole 8 x 8 images; Cin % 64; 160- or 128-channel parts; one launch below 4 GiB
    assert L.sta_conv3x3_nhwc_supported(64, 64, 64, 960, 320) and L.sta_conv3x3_nhwc_supported(64, 16, 16, 2560, 1280)
    assert L.sta_conv3x3_nhwc_supported(64, 8, 8, 1280, 1280) and L.sta_conv3x3_nhwc_supported(32, 512, 512, 128, 128)
    assert not L.sta_conv3x3_nhwc_supported(64, 64, 64, 4, 320) and not L.sta_conv3x3_nhwc_supported(64, 64, 64, 320, 4)
    assert not L.sta_conv3x3_nhwc_supported(64, 48, 48, 640, 640) and not L.sta_conv3x3_nhwc_supported(64, 12, 12, 1280, 1280)
    assert not L.sta_conv3x3_nhwc_supported(32, 512, 512, 256, 256)                     # 4.3 GB of output: the Python side splits the batch
    assert L.sta_conv3x3_packed_w_bytes(320, 320) == 320 * 320 * 9 * 2 and L.sta_conv3x3_packed_w_bytes(320, 100) == 0
    assert L.sta_conv3x3_stats_slots(64, 64) == 4 * 16 and L.sta_conv3x3_stats_slots(16, 16) == 4 and L.sta_conv3x3_stats_slots(8, 8) == 2
    assert L.sta_conv3x3_stats_slots(12, 12) == 0
    assert fused.conv3x3_work_items(64, 64, 64, 320) == 64 * 16 * 2 and fused.conv3x3_work_items(2, 16, 16, 1280) == 16
    assert fused.conv3x3_work_items(3, 8, 8, 1280) == 2 * 8 and fused.conv3x3_work_items(2, 64, 64, 128) == 32
    assert L.sta_conv3x3_nhwc(0, 0, 0, 0, 0, 0, 0, 64, 64, 64, 320, 320, 0, 1, 0) == -1 and b"null" in L.sta_last_error()
    assert L.sta_conv3x3_pack_w(8, 1, 1, 1, 1, 8, 100, 320, 1, 0) != 0 and b"Cin" in L.sta_last_error()
    # row GEMM
    assert L.sta_linear_rows_supported(262144, 320, 320) and L.sta_linear_rows_supported(4096, 2560, 1280) and L.sta_linear_rows_supported(100, 64, 128)
    assert not L.sta_linear_rows_supported(4096, 100, 320) and not L.sta_linear_rows_supported(4096, 320, 100)
    assert not L.sta_linear_rows_supported(1 << 24, 320, 320)                           # 10.7 GB of output
    assert L.sta_linear_rows_packed_w_bytes(640, 5120) == 640 * 5120 * 2
    assert L.sta_linear_rows(0, 0, 0, 0, 0, 0, 4096, 320, 320, 1, 0) == -1 and b"null" in L.sta_last_error()
    assert L.sta_linear_rows_cat(8, 8, 100, 8, 8, 0, 0, 8, 4096, 640, 320, 1, 0) != 0 and b"Ka" in L.sta_last_error()
    assert L.sta_linear_rows_stats(8, 8, 8, 0, 0, 8, 8, 100, 4096, 320, 320, 1, 0) != 0 and b"rows_per_image" in L.sta_last_error()
    assert L.sta_groupnorm_silu_nhwc_cstats(0, 0, 320, 0, 0, 0, 0, 0, 0, 2, 320, 4096, 32, 1e-5, 1, 1, 0) == -1
    assert L.sta_groupnorm_silu_nhwc_cat(8, 8, 100, 0, 8, 8, 8, 8, 2, 640, 4096, 32, 1e-5, 1, 1, 0) != 0 and b"Ca" in L.sta_last_error()
    assert L.sta_stats_finalize(0, 0, 2, 4, 320, 0) == -1
    # the product's switches exist with their production values
    assert fused.CONV3X3 and fused.LINEAR_ROWS and fused.CAT_IN_PLACE and fused.GN_STATS_FROM_PRODUCER and fused.CONV_MIN_ITEMS == 64


def test_projection_fused_forward_lds_plan():
    """Host-only rules of sta_xattn_fwd_proj (csrc/sta_xattn_proj.hip): which shapes keep every context resident in the CU's 160 KiB, which
    keep the Wq slice + the two mandatory contexts and read the local contexts from L2 (SD-v1 level 1), which are refused; and what the model
    asks (launch size, the PROJ_LL2_IN_MODEL switch). No GPU work."""
    from sta import lib, ops
    L = lib.load()
    S, LL2 = L.sta_xattn_fwd_proj_supported, L.sta_xattn_fwd_proj_locals_from_l2
    assert S(320, 8, 77, 2) and not LL2(320, 8, 77, 2)            # level 0: 30 KiB of Wq + 4 x 19 KiB
    assert S(320, 8, 77, 4) and not S(320, 8, 77, 5)              # 6 contexts fit, 7 do not
    assert S(640, 8, 77, 0) and not LL2(640, 8, 77, 0)            # level 1 without objects: 100 + 2 x 30 KiB = the whole LDS
    assert S(640, 8, 77, 2) and LL2(640, 8, 77, 2) and LL2(640, 8, 77, 4)
    # level 2: 400 KiB of Wq per head — held by the streamed-Wq kernel (2-slot LDS ring), which the model does not ask for (measured slower
    # than the GEMM + sta_xattn_fwd: profiles/r06_level2_proj.md); d = 144 (C = 1152) has no kernel
    assert S(1280, 8, 77, 2) and not LL2(1280, 8, 77, 2) and ops.proj_streams_wq(1280, 8)
    assert not S(1152, 8, 77, 2) and not ops.proj_streams_wq(1152, 8)
    assert not ops.PROJ_WQS_IN_MODEL and not ops.proj_supported(1280, 8, 77, 2, N=256, n_img=64) and ops.proj_supported(1280, 8, 77, 2)
    assert not S(640, 8, 64, 2)                                   # M <= 64 keys: refused by every forward of this family
    lib.set_option(lib.OPT_PROJ_LL2, 2)
    try:
        assert not S(640, 8, 77, 2) and not LL2(640, 8, 77, 2) and S(320, 8, 77, 2)
    finally:
        lib.set_option(lib.OPT_PROJ_LL2, 0)
    assert L.sta_xattn_fwd_proj_qfrag_supported(4, 1024, 640, 8, 77, 2) and not L.sta_xattn_fwd_proj_qfrag_supported(4, 1000, 640, 8, 77, 2)
    # the model's question: enough workgroups to fill the chip, and the switch
    assert ops.PROJ_LL2_IN_MODEL
    assert ops.proj_supported(640, 8, 77, 2, N=1024, n_img=64) and ops.proj_supported(640, 8, 77, 2, N=1024, n_img=4)
    assert not ops.proj_supported(640, 8, 77, 2, N=1024, n_img=3)                 # 192 workgroups
    ops.PROJ_LL2_IN_MODEL = False
    try:
        assert not ops.proj_supported(640, 8, 77, 2, N=1024, n_img=64) and ops.proj_supported(640, 8, 77, 2)
        assert ops.proj_supported(320, 8, 77, 2, N=4096, n_img=64)
    finally:
        ops.PROJ_LL2_IN_MODEL = True


def test_fused_row_passes_are_gated_by_what_a_launch_addresses():
    """the fragment-order chain of level 0 refuses rows * width * 2 >= 4 GiB inside the C-ABI, where no row-major fallback
    is left; the Python gate in front of the chain must say no first (205 prompts per step at 512^2 for the [rows, 1280] GEGLU output)."""
    from sta import fused
    mk = lambda prompts, n=4096: torch.empty(2 * prompts, n, 320, dtype=torch.float16, device="meta")
    assert fused.rowgemm_worthwhile(mk(64)) and fused.rowgemm_worthwhile(mk(64), 1280)
    assert fused.rowgemm_worthwhile(mk(204), 1280) and not fused.rowgemm_worthwhile(mk(205), 1280)      # 2 * 205 * 4096 * 1280 * 2 B > 4 GiB - 16
    assert fused.rows_addressable(mk(205)) and fused.rows_addressable(mk(819)) and not fused.rows_addressable(mk(820))
    assert fused.rowgemm_worthwhile(mk(51, 16384), 1280) and not fused.rowgemm_worthwhile(mk(52, 16384), 1280)  # 1024^2: N = 16384
    assert not fused.rowgemm_worthwhile(mk(7))                                       # too few rows to fill the chip: library GEMMs
