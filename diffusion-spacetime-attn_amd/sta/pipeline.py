"""Assemble the SD-v1 latent-diffusion stack around the fused cross-attention.

`build_sd_v1` gives the architecture of configs/stable-diffusion/v1-inference.yaml (UNet 859.5 M
parameters, KL-VAE decoder, 77x768 text contexts). Without a checkpoint (`ckpt=None`) the weights
are synthetic and the text encoder is the deterministic stand-in — there is no network in the build
environment; with a checkpoint the SD-v1-4 state_dict loads by name (strict=False, as the reference
does, scripts/txt2img-gpt.py:55-72)."""
import json
import os

import torch

from ldm.models.autoencoder import AutoencoderKL
from ldm.models.diffusion.ddpm import SD_V1_UNET, LatentDiffusion
from ldm.modules.diffusionmodules.openaimodel import UNetModel
from ldm.modules.encoders.modules import FrozenCLIPEmbedder, SyntheticTextEmbedder
from sta import synth

# MIOpen measures its solvers per convolution shape the first time it sees one (find mode, below): 6.5 minutes for the
# fp16 UNet + VAE at 16 prompts per step on a fresh box, 7 s when its USER find-db already holds the answers. The db is
# 30 KB of text keyed by (gfx950, MIOpen build): the one measured on an MI355X of the pool ships in sta/data/miopen_userdb
# and is used unless the caller points MIOPEN_USER_DB_PATH elsewhere. A different MIOpen build ignores the file (its
# name carries the build id) and searches as before; new shapes are searched and appended.
USER_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "miopen_userdb")


def use_shipped_miopen_db(rank=None):
    """Point MIOpen's USER find-db at a per-user (and per-rank) COPY of the shipped one. MIOpen appends newly searched shapes
    to the files of that directory, so the tracked files inside the package are never the live db (they would change on every
    run, and all ranks of a multi-GPU job would write them at once). Explicit opt-in of bench.py and the entry-point scripts;
    importing this module changes nothing in the environment. A MIOPEN_USER_DB_PATH the caller already set wins."""
    if "MIOPEN_USER_DB_PATH" in os.environ or not os.path.isdir(USER_DB):
        return os.environ.get("MIOPEN_USER_DB_PATH")
    import shutil
    if rank is None:
        rank = int(os.environ.get("LOCAL_RANK", "0"))
    import socket
    # one live copy per (host, local rank): ranks of different nodes sharing $HOME never append to the same files
    dst = os.path.join(os.path.expanduser(os.environ.get("XDG_CACHE_HOME", "~/.cache")), "sta", "miopen",
                       "%s-rank%d" % (socket.gethostname() or "host", rank))
    try:
        os.makedirs(dst, exist_ok=True)
        for name in os.listdir(USER_DB):
            if not os.path.exists(os.path.join(dst, name)):
                tmp = os.path.join(dst, ".%s.%d.tmp" % (name, os.getpid()))
                shutil.copy(os.path.join(USER_DB, name), tmp)
                os.replace(tmp, os.path.join(dst, name))          # atomic: a half-copied file never passes the exists() check
    except OSError:
        return None                      # read-only home: MIOpen searches as on a fresh box
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst


DEFAULT_CENTRES = [(0.30, 0.40), (0.70, 0.60), (0.5, 0.2), (0.25, 0.75)]


# prefixes of the checkpoint keys the sampling path consumes: a real checkpoint must provide every one of them
_REQUIRED_PREFIXES = ("model.diffusion_model.", "first_stage_model.", "cond_stage_model.")


def build_sd_v1(device="cuda", dtype=torch.float16, ckpt=None, seed=0, with_vae=True, use_checkpoint=False,
                init_weights=True, unet_overrides=None, channels_last=False, clip_tokenizer=None, real_text_encoder=None):
    """`ckpt` given: the SD-v1-4 state_dict loads by name INCLUDING the CLIP text encoder
    (`cond_stage_model.transformer.*` -> FrozenCLIPEmbedder, reference v1-inference.yaml:67-68); a key of the
    UNet / VAE decoder / text encoder that the checkpoint lacks raises (the tensors are created uninitialised).
    `ckpt=None`: synthetic weights and the deterministic text stand-in (benchmarks, tests).
    `real_text_encoder`: ranks that receive the weights by broadcast build the real (empty) encoder too."""
    cfg = dict(SD_V1_UNET, use_checkpoint=use_checkpoint)
    cfg.update(unet_overrides or {})
    # parameters are created on the meta device (no default init of 0.9 G values) and materialised
    # uninitialised on `device`; they are then filled synthetically, loaded, or received by broadcast
    with torch.device("meta"):
        unet = UNetModel(**cfg)
        vae = AutoencoderKL() if with_vae else None
    unet = unet.to(dtype).to_empty(device=device)
    if vae is not None:
        vae = vae.to(dtype).to_empty(device=device)
    if real_text_encoder is None:
        real_text_encoder = ckpt is not None
    if real_text_encoder:
        text = FrozenCLIPEmbedder(device=device, from_config=True, tokenizer_path=clip_tokenizer).to(device)
    else:
        text = SyntheticTextEmbedder().to(device)
    model = LatentDiffusion(unet_config=unet, first_stage_config=vae, cond_stage_config=text).to(device)
    if channels_last and torch.device(device).type == "cuda":
        # NHWC activations and weights for the UNet trunk: MIOpen's bf16 convolutions are NHWC kernels (the NCHW path
        # wraps each of them in two transposes), the b c h w <-> b (hw) c reshapes around the transformer blocks
        # become views and the 1x1 projections plain GEMMs. Pays only with the fused inference kernels (NHWC
        # GroupNorm in csrc/sta_unet.hip): +2.6 % images/s at 8 prompts per step; under autograd PyTorch's
        # GroupNorm makes NCHW copies of NHWC inputs and eats the gain, so callers enable it for fixed weights only.
        unet.to(memory_format=torch.channels_last)
        if vae is not None:
            # the decoder too: its 3x3 convolutions (128 / 256 / 512 channels at up to 512 x 512) then run on csrc/sta_conv.hip —
            # 32 images decode in 186 ms NCHW / 150 ms NHWC through the library, whose 128-channel 512 x 512 kernels reach 80 TFLOP/s
            vae.to(memory_format=torch.channels_last)
    if torch.device(device).type == "cuda" and os.environ.get("STA_CONV_FIND", "1") != "0":
        # Let MIOpen MEASURE its solvers per convolution shape (find mode) instead of taking the immediate-mode
        # heuristic: at the UNet's shapes the heuristic picks asm implicit-GEMM kernels where CK kernels are up to
        # 2x faster (37 vs 51 ms of convolutions per 3 UNet calls at 8 prompts per step). Costs ~30 s once per
        # process at the first call of each new shape (exclude the naive reference solvers, see bench.py).
        torch.backends.cudnn.benchmark = True
    model.eval()
    for p in model.parameters():
        p.requires_grad_(False)      # frozen: the optimisation variable is the weights tensor only (plms.py:214)
    if ckpt is not None:
        # sd-v1-4.ckpt pickles pytorch_lightning callback objects next to the state_dict: torch >= 2.6 refuses them under the
        # weights_only default. The checkpoint is the user's own trusted file (as for the reference, txt2img-gpt.py:57).
        sd = torch.load(ckpt, map_location="cpu", weights_only=False)
        sd = sd.get("state_dict", sd)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        # parameters were materialised uninitialised (to_empty): a key the checkpoint lacks would stay HBM garbage
        bad = [k for k in missing if k.startswith(_REQUIRED_PREFIXES)]
        if bad:
            raise RuntimeError("%s lacks %d tensors of the sampling path (first: %s)" % (ckpt, len(bad), ", ".join(bad[:5])))
        print("loaded %s: %d unexpected keys ignored (encoder / EMA / loss buffers)" % (ckpt, len(unexpected)))
    elif init_weights:
        if torch.device(device).type == "cuda":
            synth.device_fill_(model.model, seed)
            if vae is not None:
                synth.device_fill_(vae, seed + 1)
        else:
            synth.seeded_fill_(model.model, seed)
            if vae is not None:
                synth.seeded_fill_(vae, seed + 1)
    return model


def set_recompute(model, mode="auto", prompts_per_step=1):
    """Which blocks re-run their forward during backward (weight-optimisation epochs only).

    The reference checkpoints every ResBlock and transformer block (util.py:105-145) because its 51-call autograd
    graph would not fit a 24-48 GB GPU. One prompt at 512x512 needs 79.5 GiB of saved activations WITHOUT any
    recomputation and an MI355X has 288 GB, so the policy is sized to HBM instead:
      none — keep everything (<= 2 prompts per step: 156 GiB; +22 % images/s over `all`)
      res  — ResBlocks recompute, transformer blocks keep their activations (<= 4 prompts: 240 GiB; +8 %)
      all  — the reference's policy
      call — recompute at UNet-CALL granularity (ldm.models.diffusion.plms._CallRecompute): the forward of a tracked epoch
             is the fixed-weight path (inference kernels, hipGraph) and keeps 16 KB per call and image; backward re-runs one
             call under autograd at a time (1.1 GiB per prompt) — what lets 16-32 prompts share a step. The re-run uses an
             NHWC trunk with the differentiable fused glue ops (sta.fused.tracked, csrc/sta_unet_bwd.hip), and the last calls of a
             trajectory keep their activations instead, as many as HBM holds (PLMSSampler._calls_to_keep)
    auto = none for <= 2 prompts per step, call above. Returns the mode applied."""
    if mode == "auto":
        mode = "none" if prompts_per_step <= 2 else "call"
    model.sta_call_recompute = mode == "call"
    unet = model.model.diffusion_model
    if mode == "call" and next(unet.parameters()).is_cuda:
        # per-call recomputation re-runs a call under autograd with the differentiable fused glue ops (sta.fused.tracked),
        # whose GroupNorm works on NHWC activations: MIOpen's NHWC convolutions need no layout transposes in either
        # direction and 'b c h w -> b (h w) c' is a view
        unet.to(memory_format=torch.channels_last)
        # (the VAE decoder stays NCHW: converted as well it measured 0.80 vs 0.83-0.84 images/s at 16 prompts per step)
    from ldm.modules.diffusionmodules.openaimodel import ResBlock
    for m in unet.modules():
        if isinstance(m, ResBlock):
            m.use_checkpoint = mode in ("all", "res")
    for blk in unet.transformer_blocks():
        blk.checkpoint = mode == "all"
    return mode


def load_prompts(n=64):
    """The first 64 MS-COCO prompts with two noun chunks each (BASELINE config 4; datasets/mscoco.txt,
    mscoco.pkl) — text only, used as keys of the synthetic embedder when CLIP weights are absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "mscoco64.json")
    return json.load(open(path))[:n]


def conditionings(model, prompt, object_names, dtype=None):
    """(uc, c, [c_i]) as the reference script builds them (scripts/txt2img-gpt.py:314-321)."""
    uc = model.get_learned_conditioning([""])
    c = model.get_learned_conditioning([prompt])
    local = [model.get_learned_conditioning(["a photo of " + name]) for name in object_names]
    if dtype is not None:
        uc, c, local = uc.to(dtype), c.to(dtype), [l.to(dtype) for l in local]
    return uc, c, local
