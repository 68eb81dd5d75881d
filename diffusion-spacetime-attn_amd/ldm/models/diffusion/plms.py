"""PLMS sampler with per-object, per-timestep blend-weight optimisation.

Counterpart of the reference's ldm/models/diffusion/plms.py (PLMSSampler.sample :114-180,
plms_sampling :182-293, p_sample_plms :296-358, DCLIPLoss :21-61). Same `sample(...)` keyword
surface and the same numerics:
  * timesteps flip(1, 21, ..., 981) for S = 50; `index = S-1-i`; CFG batch [uncond, cond];
  * first step evaluates the UNet twice with the SAME weights column (pseudo improved Euler);
    Adams-Bashforth orders 2-4 afterwards; DDIM update with eta = 0;
  * weights W[K, S] start at 5/K, Adam(lr 5e-3), `opt_epochs` (default 3) full trajectories each
    followed by one optimiser step on the fidelity loss of the decoded image; the image of the last
    epoch is written to result_outputs/final{E-1}_s{seed}_index_{prompt_idx}.png.
Generalised where the reference hard-wires a constant: the width of W follows S (reference: 50),
the crop size follows the decoded image (reference: 512), the first timestep is announced to the
blocks instead of being compared with 981.

MI355X-specific: with `opt_epochs=0` (fixed weights; BASELINE config 2) the whole CFG UNet call is
captured once into a hipGraph and replayed for the 51 calls of a trajectory (sta.graphs).
"""
import math
import os

import numpy as np

import torch

from ldm.modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps
from sta import fused as _fused
from sta import prompt_state as _ps

mode = _ps.MODE


class DCLIPLoss(torch.nn.Module):
    """1 - cos(CLIP(image), CLIP(text)) with the reference's two image front-ends:
    `forward_2` (global): x7 nearest upsample then 16x16 average pool -> 224^2 for a 512^2 image (:36-44);
    `forward_3` (local crop): bilinear resize to 224^2 (:28-35, torchvision `Resize((224, 224))`).
    The CLIP model is third-party (OpenAI CLIP ViT-B/32, unpinned in the reference) and must be supplied: any
    object with `encode_image(img[1,3,224,224])` and `encode_text(tokens)`. `tokenize` is the reference's
    `clip.tokenize` (:31,:38): a callable list[str] -> token tensor; None hands the raw string to `encode_text`
    (stand-in models). See `load_clip_model` for how the entry points obtain both."""

    def __init__(self, clip_model=None, tokenize=None):
        super().__init__()
        if clip_model is None:
            raise RuntimeError("DCLIPLoss needs a CLIP model (encode_image/encode_text); none is bundled. "
                               "Pass loss_model=... to PLMSSampler, or opt_epochs=0 for fixed weights.")
        self.model = clip_model
        self.tokenize = tokenize if tokenize is not None else getattr(clip_model, "tokenize", None)
        self.upsample = torch.nn.Upsample(scale_factor=7)
        self.avg_pool = torch.nn.AvgPool2d(kernel_size=16)

    def _text(self, text, device):
        if self.tokenize is None:
            return text
        return self.tokenize([text]).to(device)                      # :31, :38

    def _loss(self, image224, text):
        fi, ft = self.model.encode_image(image224), self.model.encode_text(self._text(text, image224.device))
        return 1 - torch.nn.functional.cosine_similarity(fi, ft)

    def forward_2(self, image, text):
        return self._loss(self.avg_pool(self.upsample(image.unsqueeze(0))), text)

    def forward_3(self, image, text):
        # torchvision 0.12.0 (the reference's pin, environment_replicate.yml:10) resizes TENSORS with plain bilinear
        # interpolation, align_corners=False, no antialiasing
        img = torch.nn.functional.interpolate(image.unsqueeze(0), size=(224, 224), mode="bilinear", align_corners=False)
        return self._loss(img, text)


def load_clip_model(spec=None, device="cuda"):
    """(model, tokenize) for the fidelity loss.
      spec None           the reference's own call: `clip.load("ViT-B/32")` + `clip.tokenize` (plms.py:24,31) — needs the
                          OpenAI `clip` package and its weights on disk;
      "module:callable"   import `module`, call `callable(device)` -> model or (model, tokenize);
      path to a .pt file  `clip.load(path)`.
    Raises with the reason when nothing can be loaded — callers do this BEFORE sampling (a trajectory is 51 UNet
    calls; the reference would fail at import time, plms.py:11)."""
    import importlib
    if spec and ":" in spec and not os.path.exists(spec):
        mod, fn = spec.split(":", 1)
        got = getattr(importlib.import_module(mod), fn)(device)
        return got if isinstance(got, tuple) else (got, getattr(got, "tokenize", None))
    try:
        clip = importlib.import_module("clip")
    except ImportError as e:
        raise RuntimeError("the fidelity loss needs a CLIP model: the OpenAI `clip` package is not installed; pass "
                           "--clip module:callable (returns an object with encode_image/encode_text, optionally a "
                           "tokenizer), or --opt_epochs 0 for fixed blend weights") from e
    model, _ = clip.load(spec or "ViT-B/32", device=device)
    return model, clip.tokenize


def object_crop_box(centre, height, width, half=0.2):
    """Pixel box [y1:y2, x1:x2] of the square centre +- 0.2 clipped to the image (reference :256-270)."""
    cx, cy = centre
    x1, x2 = max(cx - half, 0), min(cx + half, 1)
    y1, y2 = max(cy - half, 0), min(cy + half, 1)
    return int(height * y1), int(height * y2), int(width * x1), int(width * x2)


class _CallRecompute(torch.autograd.Function):
    """Activation recomputation at UNet-CALL granularity (second half): the forward of a tracked epoch
    runs the whole CFG UNet call WITHOUT autograd — through the inference kernels and the captured hipGraph, exactly like a
    fixed-weight trajectory — and keeps only the call's inputs (the 16 KB latent per image and the weights column). Backward
    re-runs that one call eagerly with autograd and differentiates it w.r.t. (x, coef). The reference checkpoints every block
    instead (util.py:105-145) because 51 calls of saved activations do not fit its GPU; here one call's activations
    (~1.6 GiB per prompt at 512^2) exist only while that call is differentiated, so 32 prompts per step fit in 288 GB and the
    launch-bound eager autograd is amortised over all of them. The gradient is the recomputed call's (16-bit roundings of the
    two forward chains differ in the last bit; tests/test_modules_gpu.py holds dW to the float64 chain)."""

    @staticmethod
    def forward(ctx, fast_fn, slow_fn, t, x, coef):
        ctx.slow_fn, ctx.t = slow_fn, t
        ctx.save_for_backward(x, coef)
        with torch.no_grad():
            return fast_fn(x, t, coef).clone()          # the graph's output buffer is reused by the next replay

    @staticmethod
    def backward(ctx, grad_out):
        x, coef = ctx.saved_tensors
        need_x, need_c = ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        x_ = x.detach().requires_grad_(need_x)          # the first call of a trajectory starts from x_T: no dx chain through conv_in
        c_ = coef.detach().requires_grad_(need_c)
        gx = gc = None
        with torch.enable_grad():
            out = ctx.slow_fn(x_, ctx.t, c_)
            wrt = [t for t, need in ((x_, need_x), (c_, need_c)) if need]
            if wrt:
                grads = list(torch.autograd.grad(out, wrt, grad_out.to(out.dtype), allow_unused=True))
                gx = grads.pop(0) if need_x else None
                gc = grads.pop(0) if need_c else None
        ctx.slow_fn = None
        return None, None, None, gx, gc


class PLMSSampler(object):
    def __init__(self, model, schedule="linear", loss_model=None, opt_epochs=3, lr=0.005, weight_init=5.0,
                 local_loss_weight=5.0, use_graph=True, save_images=True, outdir="result_outputs/", loss_scale=None, keep_calls=None,
                 **kwargs):
        """`loss_scale`: the fidelity loss is multiplied by it before backward and W.grad divided by it before the Adam step.
        None = 1 for bf16 / fp32 models; for an fp16 model the power of two that brings the scaled loss to [2^15, 2^16)
        (2^12 for the synthetic CLIP stand-in's loss of ~11; the same gradients whatever the loss model's own scale is:
        tests/test_modules_gpu.py::test_fp16_small_gradients_survive_with_loss_scaling drives a loss 2^-16 of that one), and
        a tracked epoch whose W.grad comes back non-finite is re-run at 2^-8 of the scale. The backward of a tracked epoch runs through 2 x 51 UNet calls
        and the VAE decoder in the model's 16-bit type: with a CLIP loss the per-pixel gradients are ~1e-6 and smaller, i.e.
        fp16-subnormal or zero, and W.grad would lose precision or flush to 0 silently (bf16 has fp32's exponent range and needs
        nothing). A power of two scales exactly, so the unscaled W.grad equals the unscaled computation wherever that one is
        representable (the reference relies on CUDA autocast with fp32 parameters and no scaler, scripts/txt2img-gpt.py:301)."""
        super().__init__()
        self.loss_scale = loss_scale
        self.keep_calls = keep_calls      # per-call recomputation: how many trailing calls keep their activations (None = sized to HBM)
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.clip_loss_model = loss_model
        self.opt_epochs, self.lr, self.weight_init, self.local_loss_weight = opt_epochs, lr, weight_init, local_loss_weight
        self.use_graph, self.save_images, self.outdir = use_graph, save_images, outdir
        self.last_result = None
        self._graphs = None
        self._call_key = None             # (latent shape, object count) of the trajectory being sampled: keys the measured activation size

    def register_buffer(self, name, attr):
        if isinstance(attr, np.ndarray):
            attr = torch.from_numpy(attr)
        if torch.is_tensor(attr):
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, verbose=True):
        if ddim_eta != 0:
            raise ValueError("ddim_eta must be 0 for PLMS")
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        acp = self.model.alphas_cumprod
        assert acp.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        acp32 = acp.detach().to(torch.float32).cpu().numpy()          # float32 copy, as the reference registers it
        sig, a, a_prev = make_ddim_sampling_parameters(acp32, self.ddim_timesteps, eta=ddim_eta, verbose=verbose)
        # host-side float tables: indexing them never touches the device (no sync inside the step loop)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = sig, a, a_prev
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1.0 - a)

    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0.0, mask=None, x0=None, temperature=1.0, noise_dropout=0.0, score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.0,
               unconditional_conditioning=None, text_index=None, curr_text="", bboxs_curr=None, seed=None,
               prompt_idx=None, object_names=None, local_conditionings=None, **kwargs):
        """Same keywords as the reference (:114-143) plus `local_conditionings`: the K embeddings of
        "a photo of <object>" handed over in memory (the reference passes them through files)."""
        if mask is not None or x0 is not None or quantize_x0 or score_corrector is not None or noise_dropout:
            raise NotImplementedError("inpainting / quantisation / score correction are not on this path")
        if conditioning is not None and conditioning.shape[0] != batch_size:
            print("Warning: Got %d conditionings but batch-size is %d" % (conditioning.shape[0], batch_size))
        if batch_size != 1:
            raise ValueError("the spatial-temporal blocks work on one image per CFG batch (n_samples must be 1)")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if verbose:
            print("Data shape for PLMS sampling is %s" % (size,))
        self.plms_sampling(conditioning, size, x_T=x_T, temperature=temperature,
                           unconditional_guidance_scale=unconditional_guidance_scale,
                           unconditional_conditioning=unconditional_conditioning, text_index=text_index,
                           curr_text=curr_text, bboxs_curr=bboxs_curr, seed=seed, prompt_idx=prompt_idx,
                           object_names=object_names, local_conditionings=local_conditionings)
        return None

    def sample_batch(self, S, shape, conditionings, unconditional_conditionings, bboxs, object_names, local_conditionings,
                     curr_texts=None, x_T=None, unconditional_guidance_scale=7.5, eta=0.0, seed=1, prompt_indices=None,
                     verbose=False):
        """MI355X extension: I independent prompts (same number of objects) through ONE CFG batch of 2I
        per UNet call — the reference loops over prompts with n_samples = 1 (scripts/txt2img-gpt.py:305).
        Arguments are per-image lists; every image keeps its own x_T, weights W[i] and Adam state, so the
        result of image i equals `sample(...)` on prompt i alone."""
        I = len(conditionings)
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        cond = torch.cat(list(conditionings))
        uncond = torch.cat(list(unconditional_conditionings)) if isinstance(unconditional_conditionings, (list, tuple)) \
            else unconditional_conditionings.expand(I, -1, -1)
        self.plms_sampling(cond, (I, C, H, W), x_T=x_T, unconditional_guidance_scale=unconditional_guidance_scale,
                           unconditional_conditioning=uncond, text_index=0,
                           curr_text=list(curr_texts) if curr_texts is not None else [""] * I, bboxs_curr=list(bboxs), seed=seed,
                           prompt_idx=list(prompt_indices) if prompt_indices is not None else list(range(I)),
                           object_names=list(object_names), local_conditionings=list(local_conditionings), batched=True)
        return None

    # --------------------------------------------------------------------------------------------------
    def plms_sampling(self, cond, shape, x_T=None, temperature=1.0, unconditional_guidance_scale=1.0,
                      unconditional_conditioning=None, text_index=None, curr_text="", bboxs_curr=None, seed=None,
                      prompt_idx=None, object_names=None, local_conditionings=None, batched=False, **ignored):
        assert seed is not None
        bboxs_curr = [] if bboxs_curr is None else bboxs_curr
        object_names = [] if object_names is None else object_names
        device = self.model.device
        b = shape[0]
        if not batched:          # the reference's single-image call: wrap into one-element batches
            assert len(bboxs_curr) == len(object_names)
            boxes, names, texts, pidx = [bboxs_curr], [object_names], [curr_text], [prompt_idx]
            local = None if local_conditionings is None else [local_conditionings]
        else:
            boxes, names, texts, pidx, local = bboxs_curr, object_names, curr_text, prompt_idx, local_conditionings
            assert len(boxes) == len(names) == b and all(len(bx) == len(nm) for bx, nm in zip(boxes, names))
        K = len(boxes[0])
        assert all(len(bx) == K for bx in boxes), "images of a batch must have the same number of objects"
        timesteps = self.ddim_timesteps
        S = timesteps.shape[0]
        time_range = np.flip(timesteps)
        img_input = (torch.randn(shape, device=device) if x_T is None else x_T.to(device)).clone()

        # tell the 16 blocks a new prompt (batch) starts (replaces the `time == 981` test + cwd files)
        _ps.begin_prompt(local if (batched or local is None) else local[0], first_timestep=int(time_range[0]))
        block_boxes = boxes if batched else boxes[0]

        W = torch.full((b, K, S), self.weight_init / K if K else 0.0, device=device, dtype=torch.float32)   # :204-209
        W.requires_grad_(self.opt_epochs > 0 and K > 0)
        optimizer = torch.optim.Adam([W], lr=self.lr) if W.requires_grad else None
        if W.requires_grad and self.opt_epochs > 1 and self.clip_loss_model is None:      # before the first trajectory, not after it
            raise RuntimeError("opt_epochs > 0 needs a loss_model (see DCLIPLoss / load_clip_model); use opt_epochs=0 for fixed weights")

        epochs = max(self.opt_epochs, 1)
        result = {}
        for epoch in range(epochs):
            last = epoch == epochs - 1
            # Only the image of the last epoch is kept and the weights are discarded afterwards, so the
            # backward + Adam step of the last epoch cannot change any output (reference :275-288).
            track = W.requires_grad and not last
            scale_backoff = 1.0
            by_call = track and getattr(self.model, "sta_call_recompute", False)
            while True:
                # tracked epochs: eager autograd through the 51 calls, or (sta.pipeline.set_recompute mode "call") the
                # fixed-weight forward per call + one re-run of the call under autograd in backward, with the glue passes of
                # the trunk as autograd Functions over the HIP kernels (sta.fused.tracked)
                with torch.set_grad_enabled(track), _fused.tracked(by_call):
                    img = self._trajectory(img_input.clone(), cond, unconditional_conditioning, unconditional_guidance_scale,
                                           time_range, W if batched else W[0], block_boxes, text_index,
                                           graph=self.use_graph and (not track or by_call), call_recompute=by_call)
                    x_img = None
                    if self.model.first_stage_model is not None:
                        x_img = torch.clamp((self.model.decode_first_stage(img) + 1.0) / 2.0, min=0.0, max=1.0)   # :249-250
                    if track:
                        loss = sum(self._fidelity_loss(x_img[i].float(), texts[i], boxes[i], names[i]) for i in range(b))
                        optimizer.zero_grad()
                        scale = self._loss_scale(float(loss.detach())) * scale_backoff
                        (loss * scale if scale != 1.0 else loss).backward()
                        # checked whatever the scale (a backoff can land on exactly 1.0): Adam must never see a non-finite gradient
                        if not bool(torch.isfinite(W.grad).all()):
                            if scale_backoff > 2.0 ** -24:
                                scale_backoff /= 256.0         # an fp16 overflow somewhere in the backward: same epoch again, smaller scale
                                continue
                            raise FloatingPointError("the gradient of the blend weights is not finite even with the loss scale backed "
                                                     "off to %g: refusing to step the optimizer on it" % scale)
                        if scale != 1.0:
                            W.grad.div_(scale)
                        optimizer.step()
                        result.setdefault("losses", []).append(float(loss.detach()))
                break
            if last:
                result.update(x0=img.detach(), image=None if x_img is None else x_img.detach(),
                              W=(W if batched else W[0]).detach().clone())
                if self.save_images and x_img is not None:
                    for i in range(b):
                        self._save(x_img[i], epochs - 1, seed, pidx[i])
        self.last_result = result
        return None

    def _loss_scale(self, loss_value):
        if self.loss_scale is not None:
            return float(self.loss_scale)
        if next(self.model.model.parameters()).dtype != torch.float16 or not (0.0 < abs(loss_value) < float("inf")):
            return 1.0
        return float(2.0 ** min(max(math.floor(math.log2(65536.0 / abs(loss_value))), 0), 60))

    def _fidelity_loss(self, image, curr_text, bboxs_curr, object_names):
        if self.clip_loss_model is None:
            raise RuntimeError("opt_epochs > 0 needs a loss_model (see DCLIPLoss); use opt_epochs=0 for fixed weights")
        lm = self.clip_loss_model
        loss = lm.forward_2(image, curr_text)                                                       # :252
        hgt, wid = image.shape[-2], image.shape[-1]
        for centre, name in zip(bboxs_curr, object_names):
            y1, y2, x1, x2 = object_crop_box(centre, hgt, wid)
            obj = name.lower().replace("the ", "")                                                  # :266-267
            loss = loss + self.local_loss_weight * lm.forward_3(image[:, y1:y2, x1:x2], "A photo of " + obj)   # :268-273
        return loss.sum()

    def _save(self, image, epoch, seed, prompt_idx):
        from PIL import Image
        arr = (255.0 * image.detach().float().cpu().numpy().transpose(1, 2, 0)).astype(np.uint8)
        os.makedirs(self.outdir, exist_ok=True)
        Image.fromarray(arr).save(os.path.join(self.outdir, "final%d_s%d_index_%d.png" % (epoch, seed, prompt_idx)))

    # --------------------------------------------------------------------------------------------------
    def _trajectory(self, img, cond, uncond, scale, time_range, W, bboxs_curr, text_index, graph=False, call_recompute=False):
        """W: [K, S] for one image or [I, K, S] for a batch; column i of every image is used at step i."""
        S, b, device = len(time_range), img.shape[0], img.device
        self._call_key = (tuple(img.shape[1:]), int(W.shape[-2]))     # what one call's saved activations depend on, per image
        keep = self._calls_to_keep(S + 1, b) if call_recompute and torch.is_grad_enabled() else 0
        if call_recompute and torch.is_grad_enabled():
            self.last_kept_calls = keep
        eps_fn = self._make_eps_fn(cond, uncond, scale, bboxs_curr, text_index, graph, img, call_recompute, keep_last=keep, n_calls=S + 1)
        old_eps = []
        for i, step in enumerate(time_range):
            index = S - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            ts_next = torch.full((b,), int(time_range[min(i + 1, S - 1)]), device=device, dtype=torch.long)
            img, _, e_t = self._plms_update(eps_fn, img, ts, ts_next, index, old_eps, W[..., i])
            old_eps.append(e_t)
            if len(old_eps) >= 4:
                old_eps.pop(0)
        return img

    def _calls_to_keep(self, n_calls, batch):
        """Per-call recomputation re-runs every call's forward in backward; the LAST calls of a trajectory are differentiated
        first and freed first, so as many of them as HBM holds simply keep their activations (no re-run). The size of one call's
        saved activations is measured on the first kept call (none is assumed before that: the first tracked epoch of a
        sampler keeps one call); four calls' worth + 24 GiB stay free for the calls that are recomputed and the VAE backward."""
        if not torch.cuda.is_available() or self.keep_calls == 0:
            return 0
        if self.keep_calls is not None:
            return min(int(self.keep_calls), n_calls)
        per_image = getattr(self, "_call_bytes_per_image", None)
        if per_image is None or getattr(self, "_call_bytes_key", None) != self._call_key:
            return 1                          # nothing measured for this (latent shape, object count) yet: keep one call and measure it
        est = per_image * batch
        free, _ = torch.cuda.mem_get_info()
        free += torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
        return int(max(0, min(n_calls, (free - 4 * est - (24 << 30)) // max(est, 1))))

    def _make_eps_fn(self, cond, uncond, scale, bboxs_curr, text_index, graph, img, call_recompute=False, keep_last=0, n_calls=0):
        """eps(x, t, coef) with classifier-free guidance. The UNet batch is [uncond_0, cond_0, uncond_1, cond_1, ...]:
        for one image this is the reference's `cat([uc, c])` (:304-308); for a batch the pairs stay adjacent,
        which is the layout the fused kernel indexes (image-major, row 0 = uncond, row 1 = cond)."""
        if uncond is None or scale == 1.0:
            raise ValueError("the spatial-temporal path needs classifier-free guidance (scale != 1, uc given): "
                             "the blend subtracts the unconditional row (attention.py:290)")
        wdtype = next(self.model.model.parameters()).dtype
        b = img.shape[0]
        pair = lambda u, c: torch.stack([u, c], dim=1).reshape(2 * b, *u.shape[1:])
        c_in = pair(uncond.expand(b, -1, -1), cond).to(wdtype)      # built once per trajectory (the reference: per call)
        apply_fn = self.model.apply_model_extra
        if graph and img.is_cuda:
            from sta.graphs import GraphedEps
            if self._graphs is None:
                self._graphs = GraphedEps(self.model)
            apply_fn = self._graphs.bind(c_in, bboxs_curr, text_index)

        def unet(fn, x, t, coef):
            return fn(pair(x, x), text_index, pair(t, t), c_in, coef=coef, bboxs_curr=bboxs_curr)

        calls = [0]

        def eps(x, t, coef):
            k = calls[0]
            calls[0] += 1
            if call_recompute and torch.is_grad_enabled() and k >= n_calls - keep_last:
                # one of the last calls: plain autograd, its activations stay until backward reaches it (_calls_to_keep)
                before = torch.cuda.memory_allocated() if x.is_cuda else 0
                out = unet(self.model.apply_model_extra, x, t, coef)
                if x.is_cuda and k == n_calls - 1:
                    self._call_bytes_per_image = max(torch.cuda.memory_allocated() - before, 1) // b
                    self._call_bytes_key = self._call_key
            elif call_recompute and torch.is_grad_enabled():
                out = _CallRecompute.apply(lambda x_, t_, c_: unet(apply_fn, x_, t_, c_),
                                           lambda x_, t_, c_: unet(self.model.apply_model_extra, x_, t_, c_), t, x, coef)
            else:
                out = unet(apply_fn, x, t, coef)
            out = out.reshape(b, 2, *out.shape[1:])
            e_u, e_c = out[:, 0], out[:, 1]
            return e_u + scale * (e_c - e_u)
        return eps

    def _x_prev(self, x, e_t, index):
        """DDIM step with sigma = 0 (:321-338); the tables live on the host as Python floats."""
        f32 = np.float32                     # the reference evaluates these scalars in float32 on the device
        a_t, a_prev = f32(self.ddim_alphas[index]), f32(self.ddim_alphas_prev[index])
        s1m = f32(self.ddim_sqrt_one_minus_alphas[index])
        pred_x0 = (x - float(s1m) * e_t) / float(np.sqrt(a_t))
        return float(np.sqrt(a_prev)) * pred_x0 + float(np.sqrt(f32(1.0) - a_prev)) * e_t, pred_x0

    def _plms_update(self, eps_fn, x, t, t_next, index, old_eps, coef):
        e_t = eps_fn(x, t, coef)
        n = len(old_eps)
        if n == 0:      # pseudo improved Euler: second evaluation at t_next with the same coef (:341-345)
            x_mid, _ = self._x_prev(x, e_t, index)
            e_prime = (e_t + eps_fn(x_mid, t_next, coef)) / 2
        elif n == 1:    # Adams-Bashforth 2 (:348)
            e_prime = (3 * e_t - old_eps[-1]) / 2
        elif n == 2:    # Adams-Bashforth 3 (:351)
            e_prime = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
        else:           # Adams-Bashforth 4 (:354)
            e_prime = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
        x_prev, pred_x0 = self._x_prev(x, e_prime, index)
        return x_prev, pred_x0, e_t

    def p_sample_plms(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1.0, noise_dropout=0.0, score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1.0, unconditional_conditioning=None, old_eps=None, t_next=None,
                      text_index=None, coef=None, bboxs_curr=None):
        """Reference-compatible single step (:296-358) for callers that drive the loop themselves."""
        if use_original_steps or quantize_denoised or score_corrector is not None or noise_dropout:
            raise NotImplementedError("option not on the spatial-temporal path")
        eps_fn = self._make_eps_fn(c, unconditional_conditioning, unconditional_guidance_scale,
                                   [] if bboxs_curr is None else bboxs_curr, text_index, False, x)
        return self._plms_update(eps_fn, x, t, t_next, index, [] if old_eps is None else old_eps, coef)
