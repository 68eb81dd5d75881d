"""Per-prompt state shared by the 16 transformer blocks of the UNet.

The reference hands the local-prompt embeddings to the blocks through files in the cwd
(`c{i}_fix_radius_0p2_g{ID}.pt`, scripts/txt2img-gpt.py:323 -> ldm/modules/attention.py:246) and
rebuilds its per-prompt state when `time == 981` (attention.py:240). Here the sampler announces a
new prompt with :func:`begin_prompt`; blocks compare the version stamp and re-pack their K/V image
(once per prompt — K and V do not depend on the timestep). The file side channel is still honoured
when no embeddings were announced, so a caller that only writes the .pt files keeps working.
"""
import os

import torch

MODE = "fix_radius_0p2"            # attention.py:14, plms.py:19, txt2img-gpt.py:301


class _State:
    def __init__(self):
        self.version = 0
        self.local_ctx = None      # list of [1, M, Dc] tensors or None (-> file side channel)
        self.first_timestep = 981  # first DDIM timestep of a 50-step schedule (attention.py:240)


_STATE = _State()


def begin_prompt(local_ctx=None, first_timestep=None):
    """Announce a new prompt. `local_ctx`: list of K tensors [1, M, Dc] ("a photo of <object i>")."""
    _STATE.version += 1
    if local_ctx is None:
        _STATE.local_ctx = None
    else:
        _STATE.local_ctx = [c.detach() if torch.is_tensor(c) else [t.detach() for t in c] for c in local_ctx]
    if first_timestep is not None:
        _STATE.first_timestep = int(first_timestep)
    return _STATE.version


def version():
    return _STATE.version


def first_timestep():
    return _STATE.first_timestep


def local_contexts(num_objects, n_img, device, dtype):
    """The local-prompt embeddings as one [n_img, K, M, Dc] tensor on `device` (None when K == 0).

    begin_prompt() takes a list of K tensors for one image or a list of n_img such lists for a batch."""
    if num_objects == 0:
        return None
    if _STATE.local_ctx is not None:
        per_img = _STATE.local_ctx
        if n_img == 1 and (len(per_img) == 0 or torch.is_tensor(per_img[0])):
            per_img = [per_img]
        if len(per_img) != n_img or any(len(cs) != num_objects for cs in per_img):
            raise ValueError("begin_prompt() announced %s local prompts but the call has %d image(s) x %d objects"
                             % ([len(cs) for cs in per_img], n_img, num_objects))
    else:                              # drop-in fallback: the reference's cwd files (single image only)
        if n_img != 1:
            raise ValueError("the file side channel carries one image; announce batches with begin_prompt()")
        from process_id import NON_EXISTING_NAME_ID
        cs = []
        for i in range(num_objects):
            path = "c%d_%s_g%d.pt" % (i, MODE, NON_EXISTING_NAME_ID)
            if not os.path.exists(path):
                raise FileNotFoundError(
                    "no local-prompt embeddings: call sta.prompt_state.begin_prompt(local_ctx) or provide %s" % path)
            cs.append(torch.load(path, map_location="cpu"))
        per_img = [cs]
    rows = [torch.cat([c.reshape(1, c.shape[-2], c.shape[-1]) for c in cs]) for cs in per_img]
    return torch.stack(rows).to(device=device, dtype=dtype)
