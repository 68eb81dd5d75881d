"""Schedules, timestep embedding, GroupNorm32 and activation recomputation for the SD-v1 UNet.

Counterpart of the reference's ldm/modules/diffusionmodules/util.py (file:line cited per function);
only what the PLMS + spatial-temporal-attention path uses is provided.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# -- diffusion schedule -------------------------------------------------------------------------------
def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """float64 numpy betas (reference util.py:21-43). SD-v1 uses "linear" with 0.00085 / 0.0120."""
    if schedule == "linear":      # linear in sqrt(beta)
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    if schedule == "sqrt_linear":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    if schedule == "sqrt":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64) ** 0.5
    if schedule == "cosine":
        t = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        a = np.cos(t / (1 + cosine_s) * np.pi / 2) ** 2
        a = a / a[0]
        return np.clip(1 - a[1:] / a[:-1], 0, 0.999)
    raise ValueError("schedule '%s' unknown." % schedule)


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """Sub-sequence of DDPM timesteps, shifted by +1 (reference util.py:46-61): S=50 -> 1, 21, ..., 981."""
    if ddim_discr_method == "uniform":
        stride = num_ddpm_timesteps // num_ddim_timesteps
        base = np.arange(0, num_ddpm_timesteps, stride)
    elif ddim_discr_method == "quad":
        base = (np.linspace(0, np.sqrt(num_ddpm_timesteps * 0.8), num_ddim_timesteps) ** 2).astype(int)
    else:
        raise NotImplementedError('There is no ddim discretization method called "%s"' % ddim_discr_method)
    steps = base + 1
    if verbose:
        print("Selected timesteps for ddim sampler: %s" % steps)
    return steps


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """(sigmas, alphas, alphas_prev) at the selected timesteps (reference util.py:64-75)."""
    alphacums = np.asarray(alphacums)
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.concatenate([alphacums[:1], alphacums[ddim_timesteps[:-1]]])
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if verbose:
        print("ddim alphas a_t: %s; a_(t-1): %s; sigma_t (eta=%s): %s" % (alphas, alphas_prev, eta, sigmas))
    return sigmas, alphas, alphas_prev


def noise_like(shape, device, repeat=False):
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


# -- activation recomputation ---------------------------------------------------------------------------
class _Recompute(torch.autograd.Function):
    """Run `fn` without keeping activations; re-run it in backward (reference util.py:123-145).

    Output-preserving difference from the reference's CheckpointFunction: gradients are taken only
    w.r.t. the inputs/parameters that require them (the reference differentiates every block
    parameter and throws the result away, util.py:140-145).
    """

    @staticmethod
    def forward(ctx, fn, n_inputs, *args):
        ctx.fn = fn
        ctx.inputs = list(args[:n_inputs])
        ctx.params = list(args[n_inputs:])
        from sta import fused as _fused
        # same op chain as the re-run in backward (eager ops + the differentiable fused cross-attention): the
        # inference-only kernels round differently, and the gradient must belong to the activations of THIS forward
        with torch.no_grad(), _fused.held():
            return fn(*ctx.inputs)

    @staticmethod
    def backward(ctx, *grad_out):
        inputs = [x.detach().requires_grad_(x.requires_grad) for x in ctx.inputs]
        with torch.enable_grad():
            out = ctx.fn(*inputs)
        wrt = [x for x in inputs if x.requires_grad] + [p for p in ctx.params if p.requires_grad]
        grads = iter(torch.autograd.grad(out, wrt, grad_out, allow_unused=True) if wrt else ())
        g_in = [next(grads) if x.requires_grad else None for x in inputs]
        g_par = [next(grads) if p.requires_grad else None for p in ctx.params]
        ctx.inputs = ctx.params = ctx.fn = None
        return (None, None, *g_in, *g_par)


def checkpoint(func, inputs, params, flag):
    """Same call signature as the reference's `checkpoint` (util.py:105-120). Recomputation is only
    engaged when gradients are being recorded; under no_grad (plain sampling) it is a direct call."""
    inputs, params = tuple(inputs), tuple(params)
    if not flag or not torch.is_grad_enabled():
        return func(*inputs)
    if not any(t.requires_grad for t in inputs + params):
        return func(*inputs)
    return _Recompute.apply(func, len(inputs), *inputs, *params)


# -- small layers -----------------------------------------------------------------------------------------
def zero_module(module):
    """Zero all parameters (reference util.py:187-193); used for the residual-closing convs."""
    for p in module.parameters():
        p.detach().zero_()
    return module


class GroupNorm32(nn.GroupNorm):
    """GroupNorm whose statistics are evaluated in float32 whatever the activation dtype (reference
    util.py:216-218 does `super().forward(x.float()).type(x.dtype)`). On the GPU PyTorch's group_norm
    already accumulates mean/variance of 16-bit inputs in float32 and rounds once at the output, so the
    two explicit cast kernels per call (122 per UNet call) are skipped there; elsewhere the cast form is kept."""

    def forward(self, x):
        if x.is_cuda and x.dtype in (torch.bfloat16, torch.float16):
            return F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        w = None if self.weight is None else self.weight.float()
        b = None if self.bias is None else self.bias.float()
        return F.group_norm(x.float(), self.num_groups, w, b, self.eps).type(x.dtype)


def normalization(channels):
    return GroupNorm32(32, channels)


def Normalize(in_channels):
    """The transformer/VAE flavour: 32 groups, eps 1e-6 (reference attention.py:74-75)."""
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """Sinusoidal embedding [cos | sin] (reference util.py:151-172)."""
    if repeat_only:
        return timesteps[:, None].repeat(1, dim)
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb
