"""hipGraph replay of the classifier-free-guidance UNet call (fixed-weights sampling, no autograd).

One UNet call of SD-v1 at batch 2 is ~1000 small launches; issued eagerly from Python it is
launch-bound on MI355X. All shapes are static inside a trajectory, so the call is captured once per
(input shape, dtype, number of objects) and replayed for the 51 calls of every trajectory.

What makes the capture reusable across prompts: nothing per-prompt is baked into it except device
ADDRESSES. The per-prompt data — packed K/V images, disc masks — live in buffers that each
transformer block allocates once per shape and refills in place (BasicTransformerBlock.prepare_prompt);
the per-step data — latent, timestep, weights column — are copied into static input tensors before
each replay.
"""
import torch

from sta import prompt_state as _ps


class _Entry:
    __slots__ = ("graph", "x", "t", "coef", "out", "blocks", "version", "centres")


class GraphedEps:
    def __init__(self, model, warmup=2):
        self.model = model
        self.warmup = warmup
        self._entries = {}

    def _unet(self):
        return self.model.model.diffusion_model

    def bind(self, c_in, bboxs_curr, text_index):
        """Return an apply_model_extra-compatible callable for this prompt (context + objects)."""
        n_img = c_in.shape[0] // 2
        from ldm.modules.attention import BasicTransformerBlock
        centres = BasicTransformerBlock._per_image_boxes(bboxs_curr, n_img)
        K = len(centres[0])
        boxes = [list(map(list, c)) for c in centres] if n_img > 1 else [list(b) for b in centres[0]]

        def apply(x_in, text_index_, t_in, c_in_, coef=None, bboxs_curr=None):
            key = (tuple(x_in.shape), x_in.dtype, K)
            ent = self._entries.get(key)
            if ent is None:
                ent = self._capture(key, x_in, t_in, c_in, coef, boxes, K, text_index)
                ent.centres = centres
            elif ent.version != _ps.version() or ent.centres != centres:
                # new prompt: refill every block's K/V image and masks in place, outside the graph
                for blk, n in ent.blocks:
                    blk.prepare_prompt(n, c_in, boxes)
                ent.version, ent.centres = _ps.version(), centres
            ent.x.copy_(x_in)
            ent.t.copy_(t_in)
            if K:
                ent.coef.copy_(coef.detach())
            ent.graph.replay()
            return ent.out
        return apply

    def _capture(self, key, x_in, t_in, c_in, coef, centres, K, text_index):
        ent = _Entry()
        ent.x, ent.t = x_in.clone(), t_in.clone()
        ent.coef = coef.detach().to(torch.float32).clone() if K else None
        fn = lambda: self.model.apply_model_extra(ent.x, text_index, ent.t, c_in, coef=ent.coef, bboxs_curr=centres)
        side = torch.cuda.Stream(device=x_in.device)
        side.wait_stream(torch.cuda.current_stream(x_in.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):      # builds block caches, MIOpen/hipBLASLt plans, kernel attributes
                fn()
        torch.cuda.current_stream(x_in.device).wait_stream(side)
        ent.blocks = [(blk, blk._last_n) for blk in self._unet().transformer_blocks()]
        ent.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(ent.graph):
            ent.out = fn()
        ent.version, ent.centres = _ps.version(), None
        self._entries[key] = ent
        return ent
