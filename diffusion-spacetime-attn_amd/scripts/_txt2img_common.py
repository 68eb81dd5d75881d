"""Shared body of scripts/txt2img-{gpt,mscoco,vsr}.py — the reference's three entry points differ only
in the dataset they read (scripts/txt2img-gpt.py vs -mscoco.py vs -vsr.py: lines 255-261).

Kept from the reference CLI (txt2img-gpt.py:105-247): --plms --ddim_steps --H --W --C --f --n_samples --scale
--ddim_eta --fixed_code --config --ckpt --precision --outdir --seed --process_id (+ the flags it parses and
ignores, accepted for compatibility). Added: --layout (JSON replacing the layout-predictor call),
--dataset (path override), --opt_epochs (0 = fixed weights), --limit/--start, --dtype, --synthetic,
--clip (the fidelity-loss model; checked BEFORE sampling), --clip_tokenizer (vocabulary of the text encoder).
With torch.distributed.run the prompts are sharded round-robin over the ranks (one GPU each).
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
if PKG not in sys.path:
    sys.path.insert(0, PKG)

import torch  # noqa: E402


def build_parser(default_dataset):
    p = argparse.ArgumentParser()
    p.add_argument("--prompt", type=str, nargs="?", default="a painting of a virus monster playing guitar",
                   help="parsed and ignored, as in the reference (prompts come from the dataset)")
    p.add_argument("--outdir", type=str, nargs="?", default="outputs/notuse")
    p.add_argument("--skip_grid", action="store_true")
    p.add_argument("--skip_save", action="store_true")
    p.add_argument("--ddim_steps", type=int, default=50)
    p.add_argument("--plms", action="store_true")
    p.add_argument("--dpm_solver", action="store_true")
    p.add_argument("--laion400m", action="store_true")
    p.add_argument("--fixed_code", action="store_true")
    p.add_argument("--ddim_eta", type=float, default=0.0)
    p.add_argument("--n_iter", type=int, default=2)
    p.add_argument("--H", type=int, default=512)
    p.add_argument("--W", type=int, default=512)
    p.add_argument("--C", type=int, default=4)
    p.add_argument("--f", type=int, default=8)
    p.add_argument("--n_samples", type=int, default=1)
    p.add_argument("--n_rows", type=int, default=0)
    p.add_argument("--scale", type=float, default=7.5)
    p.add_argument("--from-file", type=str)
    p.add_argument("--config", type=str, default="configs/stable-diffusion/v1-inference.yaml")
    p.add_argument("--ckpt", type=str, default="models/ldm/stable-diffusion-v1/model.ckpt")
    p.add_argument("--seed", type=int, default=42, help="parsed and ignored: the reference uses seed = 1 for every prompt")
    p.add_argument("--process_id", type=int, default=0)
    p.add_argument("--precision", type=str, choices=["full", "autocast"], default="autocast")
    # additions
    p.add_argument("--dataset", type=str, default=default_dataset)
    p.add_argument("--layout", type=str, default=None, help="JSON {prompt | index: {object: [x, y]}}")
    p.add_argument("--opt_epochs", type=int, default=3, help="weight-optimisation epochs (reference: 3; 0 = fixed weights)")
    p.add_argument("--start", type=int, default=0)
    p.add_argument("--limit", type=int, default=500)
    p.add_argument("--dtype", type=str, choices=["bf16", "fp16"], default="fp16",
                   help="fp16 = the reference's autocast type and the one within 1e-3 on the attention maps (DESIGN.md section 2)")
    p.add_argument("--clip", type=str, default=None,
                   help="fidelity-loss model for --opt_epochs > 0: 'module:callable' (called with the device, returns a model with "
                        "encode_image/encode_text or (model, tokenize)), a CLIP .pt path, or 'synthetic' (frozen stand-in, for "
                        "timing the gradient path). Default: clip.load('ViT-B/32') as the reference (plms.py:24)")
    p.add_argument("--fp8", action="store_true",
                   help="store the Linear weights of the transformer blocks as OCP e4m3 with per-channel scales (BASELINE configs[4]): half "
                        "the weight memory; their GEMMs become hipBLASLt's row-scaled e4m3 GEMMs behind a per-row activation quantiser, "
                        "which measures SLOWER than 16 bit here (a memory option, not a rate option: DESIGN.md section 5); fixed blend "
                        "weights only (--opt_epochs 0)")
    p.add_argument("--clip_tokenizer", type=str, default=None, help="directory with the CLIP tokenizer files (with --ckpt)")
    p.add_argument("--synthetic", action="store_true", help="synthetic weights/text embeddings when no checkpoint is available")
    p.add_argument("--batch_prompts", type=int, default=1,
                   help="sample up to this many prompts with the same object count together (one CFG batch of 2I per UNet call)")
    return p


def run(kind, default_dataset):
    opt = build_parser(default_dataset).parse_args()
    if not opt.plms:
        raise SystemExit("only --plms works with the spatial-temporal UNet (DDIM/DPM-Solver call apply_model with the "
                         "wrong positional arguments in the reference, ddim.py:172-177 vs ddpm.py:1420)")
    if opt.n_samples != 1:
        raise SystemExit("--n_samples must be 1 (the blocks reshape to the CFG batch of 2, attention.py:282)")
    from ldm.models.diffusion.plms import PLMSSampler
    from sta import datasets, parallel
    from sta.pipeline import build_sd_v1, conditionings, use_shipped_miopen_db

    rank, world, local = parallel.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("a GPU is required (the fused cross-attention has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = torch.bfloat16 if opt.dtype == "bf16" else torch.float16
    use_shipped_miopen_db(local)      # per-rank copy of the shipped MIOpen find-db; a different MIOpen build ignores it

    prompts = datasets.load_prompts(opt.dataset, kind, opt.limit)
    layouts = datasets.load_layouts(opt.layout) if opt.layout else None
    ckpt = opt.ckpt if (os.path.exists(opt.ckpt) and not opt.synthetic) else None
    if ckpt is None and not opt.synthetic:
        raise SystemExit("checkpoint %s not found (pass --synthetic to run with synthetic weights)" % opt.ckpt)
    loss_model = None
    if opt.opt_epochs > 1:        # the sampler evaluates the loss only when an epoch is tracked (opt_epochs > 1); fail here, not after the first 51-call trajectory
        from ldm.models.diffusion.plms import DCLIPLoss, load_clip_model
        if opt.clip == "synthetic":
            from sta.synth import SyntheticCLIP
            loss_model = DCLIPLoss(SyntheticCLIP().to(dev))
        else:
            try:
                loss_model = DCLIPLoss(*load_clip_model(opt.clip, dev))
            except Exception as e:
                raise SystemExit("--opt_epochs %d: %s" % (opt.opt_epochs, e))
    model = build_sd_v1(dev, dtype, ckpt=ckpt if rank == 0 else None, init_weights=(rank == 0), use_checkpoint=opt.opt_epochs > 1,
                        clip_tokenizer=opt.clip_tokenizer, real_text_encoder=ckpt is not None)
    if opt.opt_epochs > 1 and torch.device(dev).type == "cuda" and opt.H * opt.W <= 512 * 512:
        from sta.pipeline import set_recompute
        set_recompute(model, "auto", max(opt.batch_prompts, 1))      # sized to 288 GB of HBM at 512x512; larger images keep the reference's policy
    parallel.broadcast_module_(model)                                   # one RCCL broadcast of the frozen weights
    if opt.fp8:
        if opt.opt_epochs > 0:
            raise SystemExit("--fp8 is an inference option: use --opt_epochs 0")
        from sta import fp8
        n, before, after = fp8.convert_transformer_linears_(model.model.diffusion_model)
        print("[rank %d] %d Linear layers -> e4m3: %.2f GB -> %.2f GB" % (rank, n, before / 1e9, after / 1e9))
    sampler = PLMSSampler(model, opt_epochs=opt.opt_epochs, loss_model=loss_model)
    os.makedirs(opt.outdir, exist_ok=True)

    seed = 1                                                            # txt2img-gpt.py:304
    shape = [opt.C, opt.H // opt.f, opt.W // opt.f]
    todo = list(enumerate(prompts))[opt.start: opt.start + 500]
    mine = [todo[j] for j in parallel.shard_indices(len(todo), rank, world)]

    def run_one(prompt_idx, prompt, layout):
        torch.manual_seed(seed)                                         # seed_everything(seed), :306
        print("[rank %d] Start inference for %dth prompt: %s" % (rank, prompt_idx, prompt))
        names = list(layout.keys())
        uc, c, local_c = conditionings(model, prompt, names, dtype)
        x_T = torch.randn([opt.n_samples, *shape], device=dev) if opt.fixed_code else None
        sampler.sample(S=opt.ddim_steps, conditioning=c, batch_size=opt.n_samples, shape=shape, verbose=False,
                       unconditional_guidance_scale=opt.scale, unconditional_conditioning=uc, eta=opt.ddim_eta, x_T=x_T,
                       text_index=0, curr_text=prompt, bboxs_curr=[layout[n] for n in names], seed=seed,
                       prompt_idx=prompt_idx, object_names=names, local_conditionings=local_c)

    def run_group(group):
        """Prompts with the same number of objects share one CFG batch; every image keeps the reference's
        per-prompt start: seed_everything(1) then randn, i.e. the same x_T for each."""
        torch.manual_seed(seed)
        x1 = torch.randn([1, *shape], device=dev)
        conds = [conditionings(model, p, list(l.keys()), dtype) for _, p, l in group]
        print("[rank %d] Start inference for prompts %s" % (rank, [i for i, _, _ in group]))
        sampler.sample_batch(S=opt.ddim_steps, shape=shape, conditionings=[c[1] for c in conds],
                             unconditional_conditionings=[c[0] for c in conds],
                             bboxs=[[l[n] for n in l] for _, _, l in group], object_names=[list(l.keys()) for _, _, l in group],
                             local_conditionings=[c[2] for c in conds], curr_texts=[p for _, p, _ in group],
                             x_T=x1.expand(len(group), -1, -1, -1), unconditional_guidance_scale=opt.scale, eta=opt.ddim_eta,
                             seed=seed, prompt_indices=[i for i, _, _ in group])

    items = [(i, p, datasets.layout_for(layouts, p, i) or {}) for i, p in mine]
    if opt.batch_prompts <= 1:
        for i, p, l in items:
            run_one(i, p, l)
    else:
        by_k = {}
        for it in items:
            by_k.setdefault(len(it[2]), []).append(it)
        for k in sorted(by_k):
            g = by_k[k]
            for a in range(0, len(g), opt.batch_prompts):
                run_group(g[a:a + opt.batch_prompts])
    parallel.barrier()
