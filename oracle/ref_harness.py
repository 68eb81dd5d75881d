"""ORACLE tooling — imports the REFERENCE's own Python modules on CPU (build container only).

/root/reference never travels to the GPU box, so nothing under tests/, bench.py or smoke() imports
this file; it is used by oracle/gen_golden.py to produce tests/golden/*.npz and by the optional
`-m "not gpu"` cross-checks that skip themselves when /root/reference is absent.

What has to be faked to import the hot-path modules:
  * `torchvision`  — only used by the debug `plot` (attention.py:217-221) and a Resize inside the
                     CLIP loss (plms.py:17,27);
  * `omegaconf.listconfig.ListConfig` — imported inside UNetModel.__init__ (openaimodel.py:476);
  * `clip`         — plms.py:11, only touched by DCLIPLoss.__init__ / forward_2 / forward_3.
These are inert module objects placed in sys.modules; they provide no behaviour of the reference.
For the loss front-end golden (G8, `reference_loss_env`) two of them carry behaviour of the PINNED third-party
packages, not of the reference: `clip.load` hands back the frozen stand-in model sta.synth.SyntheticCLIP (the real
OpenAI CLIP is unpinned and its weights are not available: the golden pins everything AROUND it — x7 upsample +
16x16 average pool, crop + resize, 1 - cosine), and `torchvision.transforms.Resize` is restated as torchvision
0.12.0 (environment_replicate.yml:10) implements it for tensors: bilinear, align_corners=False, NO antialiasing.
`Tensor.cuda()` / `.to("cuda")` are made no-ops there (device placement only; the box has no GPU).
The reference reads `uncond_fix_radius_0p2_g0.pt` and `c{i}_fix_radius_0p2_g0.pt` relative to the
cwd (attention.py:234,246), so everything runs inside a scratch directory prepared here.
"""
import contextlib
import os
import sys
import tempfile
import types

import torch

REF_ROOT = "/root/reference"
SD = os.path.join(REF_ROOT, "attention_optimization", "stable-diffusion")
UNCOND_PT = os.path.join(SD, "uncond_fix_radius_0p2_g0.pt")


def available():
    return os.path.isdir(SD)


def _install_stubs():
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tvt.Resize = lambda *a, **k: torch.nn.Identity()
        tv.transforms = tvt
        tv.io = types.SimpleNamespace(write_png=lambda *a, **k: None)
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")
        lc = types.ModuleType("omegaconf.listconfig")

        class ListConfig(list):
            pass

        lc.ListConfig = ListConfig
        oc.listconfig = lc
        sys.modules["omegaconf"] = oc
        sys.modules["omegaconf.listconfig"] = lc
    if "clip" not in sys.modules:
        sys.modules["clip"] = types.ModuleType("clip")


class _Tokens:
    """What the `clip.tokenize` stand-in returns: remembers the text (SyntheticCLIP.encode_text hashes str(x))."""

    def __init__(self, texts):
        self.text = texts[0]

    def to(self, *a, **k):
        return self

    def __str__(self):
        return self.text


@contextlib.contextmanager
def reference_loss_env(clip_model):
    """The reference's DCLIPLoss (plms.py:21-61) runnable on CPU around `clip_model`. Yields the reference's plms module."""
    import torch.nn.functional as F

    class Resize(torch.nn.Module):            # torchvision 0.12.0, tensor input: F.interpolate(bilinear, align_corners=False)
        def __init__(self, size):
            super().__init__()
            self.size = tuple(size)

        def forward(self, img):
            return F.interpolate(img.unsqueeze(0).float(), size=self.size, mode="bilinear", align_corners=False).squeeze(0)

    with reference_env() as ref:
        tvt = sys.modules["torchvision.transforms"]
        clip = sys.modules["clip"]
        saved = (tvt.Resize, getattr(clip, "load", None), getattr(clip, "tokenize", None), torch.Tensor.cuda)
        tvt.Resize = Resize
        ref.plms.transforms.Resize = Resize
        clip.load = lambda name, device=None, **k: (clip_model, None)
        clip.tokenize = lambda texts: _Tokens(texts)
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            yield ref.plms
        finally:
            tvt.Resize, clip.load, clip.tokenize, torch.Tensor.cuda = saved


def load_uncond():
    """The reference's only tensor fixture: CLIP-L/14 embedding of "" [1,77,768] fp32 (saved from CUDA)."""
    return torch.load(UNCOND_PT, map_location="cpu").float().contiguous()


@contextlib.contextmanager
def reference_env(local_ctx=()):
    """Scratch cwd holding the CPU-remapped uncond fixture and the c{i} side-channel files; the
    reference's `ldm` package importable. Yields a namespace with the reference modules."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % SD)
    _install_stubs()
    saved_cwd, saved_path = os.getcwd(), list(sys.path)
    saved_mods = {k: v for k, v in sys.modules.items() if k == "ldm" or k.startswith("ldm.") or k == "process_id"}
    for k in saved_mods:
        del sys.modules[k]
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        # our own package also has a top-level `ldm`; a regular package beats the reference's
        # namespace package whatever the path order, so it must be off sys.path while the reference runs
        ours = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusion-spacetime-attn_amd")
        sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != ours]
        sys.path.insert(0, SD)
        try:
            torch.save(load_uncond(), "uncond_fix_radius_0p2_g0.pt")
            for i, c in enumerate(local_ctx):
                torch.save(c.clone(), "c%d_fix_radius_0p2_g0.pt" % i)
            import ldm.modules.attention as ref_attention
            import ldm.modules.diffusionmodules.openaimodel as ref_unet
            import ldm.modules.diffusionmodules.util as ref_util
            import ldm.models.diffusion.plms as ref_plms
            assert ref_attention.__file__.startswith(SD), ref_attention.__file__
            yield types.SimpleNamespace(attention=ref_attention, unet=ref_unet, util=ref_util, plms=ref_plms, tmp=tmp)
        finally:
            os.chdir(saved_cwd)
            sys.path[:] = saved_path
            for k in [k for k in sys.modules if k == "ldm" or k.startswith("ldm.") or k == "process_id"]:
                del sys.modules[k]
            sys.modules.update(saved_mods)


class FakeLatentDiffusion:
    """The ten lines of LatentDiffusion that PLMSSampler.make_schedule / p_sample_plms touch
    (ddpm.py:117-169 register_schedule, :891-906 apply_model_extra), around a reference UNet."""

    def __init__(self, ref, unet, timesteps=1000, linear_start=0.00085, linear_end=0.0120):
        import numpy as np
        betas = ref.util.make_beta_schedule("linear", timesteps, linear_start=linear_start, linear_end=linear_end)
        alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        self.betas, self.alphas_cumprod, self.alphas_cumprod_prev = f32(betas), f32(alphas_cumprod), f32(alphas_cumprod_prev)
        self.num_timesteps = timesteps
        self.device = torch.device("cpu")
        self.unet = unet

    def apply_model_extra(self, x_noisy, text_index, t, cond, return_ids=False, coef=None, bboxs_curr=None):
        return self.unet(x_noisy, text_index, t, context=cond, coef=coef, bboxs_curr=bboxs_curr)


def make_ref_sampler(ref, model):
    """PLMSSampler without its CLIP loss model (object.__new__ skips __init__, plms.py:66-72)."""
    s = object.__new__(ref.plms.PLMSSampler)
    s.model = model
    s.ddpm_num_timesteps = model.num_timesteps
    s.schedule = "linear"
    s.register_buffer = lambda name, attr: setattr(s, name, attr)   # CPU instead of forced "cuda" (plms.py:75-79)
    return s
