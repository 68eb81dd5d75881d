"""A/B of the projection-fused forward (sta_xattn_fwd_proj) against the GEMM + sta_xattn_fwd it replaces,
interleaved in one process (HIP events on the launch stream). SURVEY.md section 8f-1 / VERDICT r01 item 2."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import lib, ops  # noqa: E402

CENTRES = [(0.30, 0.40), (0.70, 0.60), (0.5, 0.2), (0.25, 0.75)]


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def timed_cold(fn, iters, scratch):
    """Every call behind a 512 MiB fill (untimed): the operands of the call come from HBM, not from the 256 MiB Infinity Cache,
    as they do inside a UNet call (profiles/r04_insitu_vs_warm.md). One event pair per call, raw times."""
    for _ in range(2):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in ev:
        scratch.add_(1)
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cold", action="store_true", help="flush the caches before every call (median of per-call event pairs)")
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--C", type=int, default=320)
    ap.add_argument("--K", type=int, default=2)
    ap.add_argument("--imgs", type=int, default=16)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--waves", type=int, nargs="*", default=[8])      # accepted for old command lines, unused
    ap.add_argument("--only", default=None, help="run only this arm (for rocprofv3): proj | gemm | attn")
    ap.add_argument("--opt", nargs="*", default=[], help="library options for the whole run, e.g. OPT_PROJ_LL2=3")
    a = ap.parse_args()
    for kv in a.opt:
        k_, v_ = kv.split("=")
        lib.set_option(getattr(lib, k_), int(v_))
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    dev, heads, M, I, N, C, K = "cuda", 8, 77, a.imgs, a.N, a.C, a.K
    g = torch.Generator().manual_seed(0)
    y = torch.randn(2 * I, N, C, generator=g).to(dt).to(dev)
    wq = (torch.randn(C, C, generator=g) / C ** 0.5).to(dt).to(dev)
    k = (torch.randn(I * (K + 2), M, C, generator=g) * 0.78).to(dt).to(dev)
    v = torch.randn(I * (K + 2), M, C, generator=g).to(dt).to(dev)
    mask = ops.disc_mask_bits(CENTRES[:K], int(N ** 0.5)).to(dev).repeat(I, 1)
    coef = torch.full((I, K), 5.0 / max(K, 1), device=dev)
    scale = (C // heads) ** -0.5
    packed, packed_p, wqf = ops.pack_kv(k, v, heads, n_img=I), ops.pack_kv_proj(k, v, heads, n_img=I), ops.pack_wq(wq, heads)
    q = torch.nn.functional.linear(y, wq)
    arms = {"gemm": lambda: torch.nn.functional.linear(y, wq),
            "attn": lambda: ops.xattn_forward(q, packed, mask, coef, scale),
            "gemm+attn": lambda: ops.xattn_forward(torch.nn.functional.linear(y, wq), packed, mask, coef, scale)}
    def proj():
        lib.set_option(lib.OPT_PROJ_PAIR, 2)
        r = ops.xattn_forward_proj(y, wqf, packed_p, mask, coef, scale)
        lib.set_option(lib.OPT_PROJ_PAIR, 0)
        return r
    arms["proj"] = proj                     # one head per workgroup (sta_xattn_proj.hip)
    def pair():
        lib.set_option(lib.OPT_PROJ_PAIR, 1)
        r = ops.xattn_forward_proj(y, wqf, packed_p, mask, coef, scale)
        lib.set_option(lib.OPT_PROJ_PAIR, 0)
        return r
    arms["pair"] = pair                     # a head pair per workgroup (sta_xattn_proj3.hip)
    if N % 16 == 0:
        yq = ops.to_qfrag(y)
        from sta import fused
        xs, fs = torch.randn_like(y), torch.randn_like(y)
        lw, lb = torch.ones(C, device=dev, dtype=dt), torch.zeros(C, device=dev, dtype=dt)
        def pairq():
            lib.set_option(lib.OPT_PROJ_PAIR, 1)
            r = ops.xattn_forward_proj(yq, wqf, packed_p, mask, coef, scale, qfrag=True)
            lib.set_option(lib.OPT_PROJ_PAIR, 0)
            return r
        arms["pairq"] = pairq               # the same kernel reading y in query-fragment order
        if C == 320:
            def pairqo():
                lib.set_option(lib.OPT_PROJ_PAIR, 1)
                r = ops.xattn_forward_proj(yq, wqf, packed_p, mask, coef, scale, qfrag=True, ofrag=True)
                lib.set_option(lib.OPT_PROJ_PAIR, 0)
                return r
            arms["pairqo"] = pairqo         # ... and writing its output in out-fragment order
            wo = (torch.randn(C, C, generator=g) / C ** 0.5).to(dt).to(dev)
            bo = torch.zeros(C, device=dev, dtype=dt)
            wof = fused.pack_to_out_weight(wo, heads)
            blo = ops.to_ofrag(y)
            arms["toln_fused"] = lambda: fused.to_out_add_layernorm_ofrag(xs, blo, wof, bo, lw, lb, 1e-5, heads)      # to_out + residual + LayerNorm, one pass
            arms["toln_gemm+ln"] = lambda: fused.add_layernorm(xs, torch.nn.functional.linear(y, wo, bo), None, lw, lb, 1e-5)   # what it replaces
        if C == 320:
            w1 = (torch.randn(2560, C, generator=g) / C ** 0.5).to(dt).to(dev)
            b1 = torch.zeros(2560, device=dev, dtype=dt)
            w1f = fused.pack_geglu_weight(w1)
            arms["ff1_fused"] = lambda: fused.ff_geglu_qfrag(yq, w1f, b1, 1280)                              # GEGLU projection + gelu * mul, one pass
            arms["ff1_gemm+geglu"] = lambda: fused.geglu(torch.nn.functional.linear(y, w1, b1))              # what it replaces
            w2 = (torch.randn(C, 1280, generator=g) / 1280 ** 0.5).to(dt).to(dev)
            b2 = torch.zeros(C, device=dev, dtype=dt)
            w2f = fused.pack_ff_out_weight(w2)
            hrm = fused.ff_geglu_qfrag(yq, w1f, b1, 1280)
            hfr = fused.ff_geglu_qfrag(yq, w1f, b1, 1280, h_frag=True)
            arms["ff1_fused_hfrag"] = lambda: fused.ff_geglu_qfrag(yq, w1f, b1, 1280, h_frag=True)
            arms["ff2_fused"] = lambda: fused.ff_out_res_hfrag(xs, hfr, w2f, b2)                              # output Linear + residual, one pass
            arms["ff2_gemm+add"] = lambda: torch.nn.functional.linear(hrm, w2, b2) + xs                       # what it replaces
        arms["ln"] = lambda: fused.add_layernorm(xs, fs, None, lw, lb, 1e-5)                     # the producer pass, row-major y
        arms["lnq"] = lambda: fused.add_layernorm(xs, fs, None, lw, lb, 1e-5, qfrag=True)        # ... query-fragment order
    if a.only:
        arms = {k_: f for k_, f in arms.items() if any(k_.startswith(o) for o in a.only.split(","))}
    res = {n: [] for n in arms}
    scratch = torch.zeros(512 << 20, dtype=torch.uint8, device=dev) if a.cold else None
    for _ in range(a.rounds):
        for n, f in arms.items():
            res[n].append(round(timed_cold(f, min(a.iters, 30), scratch) if a.cold else timed(f, a.iters), 2))
    f_attn = I * 4.0 * M * C * N * (K + 2)
    f_proj = I * 2.0 * 2 * N * C * C
    byts = I * (8.0 * N * C + 4.0 * (K + 2) * M * C + K * N) + 2.0 * C * C
    out = {"N": N, "C": C, "K": K, "imgs": I, "dtype": a.dtype, "cold": bool(a.cold), "opt": a.opt, "us": res, "attn_gflop": f_attn / 1e9, "proj_gflop": f_proj / 1e9, "mbytes": byts / 1e6}
    for n, v_ in res.items():
        if n.startswith("proj") or n.startswith("pair"):
            us = min(v_)
            out[n + "_tflops"] = round((f_attn + f_proj) / us / 1e6, 1)
            out[n + "_gbps"] = round(byts / us / 1e3, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
