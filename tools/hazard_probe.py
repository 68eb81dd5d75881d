"""Measured hazard table of gfx950 for the instruction pairs the level-0 kernels live on.

hipcc pads the hazards its recogniser knows; profiles/r04_level0.md section 5 found a pair it pads too little (an MFMA's C operand
overwritten by a vector instruction two wait states later).  This tool measures, on the GPU, how many wait states each pair really
needs, by what kind of instruction the states may be filled, and for which MFMA shapes — the table `sta/isa_lint.py` enforces
on the compiled kernels.

Every probe is one `asm volatile` statement with literal registers (nothing for the compiler to schedule or pad):

    load operands -> settle -> [first instruction] -> k fillers -> [second instruction] -> settle -> store result registers

and is compared bit for bit with the same probe at 32 wait states of s_nop.  Fillers: s_nop 0, an independent v_mov_b32, an
independent ds_read_b32, an s_mov_b32.  Usage (GPU box):  python tools/hazard_probe.py [--out gpurun_out/hazard_probe.txt]
"""
import argparse
import ctypes
import os
import random
import re
import struct
import subprocess
import sys

# name: (mnemonic, A regs, B regs, C/D regs, passes hipcc assumes)
MFMAS = {
    "bf16_16x16x32": ("v_mfma_f32_16x16x32_bf16", 4, 4, 4, 4),
    "f16_16x16x32": ("v_mfma_f32_16x16x32_f16", 4, 4, 4, 4),
    "bf16_16x16x16": ("v_mfma_f32_16x16x16_bf16", 2, 2, 4, 4),
    "f16_16x16x16": ("v_mfma_f32_16x16x16_f16", 2, 2, 4, 4),
    "bf16_32x32x16": ("v_mfma_f32_32x32x16_bf16", 4, 4, 16, 8),
    "fp8_16x16x128": ("v_mfma_f32_16x16x128_f8f6f4", 8, 8, 4, 8),
}
A0, B0, C0, D0, E0, R0 = 10, 18, 30, 50, 70, 110      # first registers of A, B, C, D, the second MFMA's D, the result block
JUNK, TMP, F_V, F_VS, F_DS, F_DSA = 100, 101, 102, 103, 104, 105
FILLERS = {"nop": "s_nop 0", "valu": "v_mov_b32 v%d, v%d" % (F_V, F_VS), "ds": "ds_read_b32 v%d, v%d" % (F_DS, F_DSA), "salu": "s_mov_b32 s40, 0"}
# independent MFMAs as fillers (they serialise on the matrix pipe: one of them is worth several issue slots to a following MFMA)
FILLERS["mfma32"] = "v_mfma_f32_16x16x32_bf16 v[194:197], v[10:13], v[18:21], v[194:197]"
FILLERS["mfma16"] = "v_mfma_f32_16x16x16_bf16 v[198:201], v[10:11], v[18:19], v[198:201]"
# round 6: transcendental and packed-fp32 vector instructions as fillers (the LDS-resident backward's softmax sits between an MFMA and
# the vector instruction that reads its result: profiles/r06_bwd.md)
FILLERS["trans"] = "v_exp_f32 v106, v%d" % F_VS
FILLERS["pkf32"] = "v_pk_mul_f32 v[106:107], v[%d:%d], v[%d:%d]" % (F_V, F_VS, F_V, F_VS)
GOLD = "s_nop 15\ns_nop 15"


def rng(first, n):
    return "v[%d:%d]" % (first, first + n - 1)


def mfma(name, d, a, b, c):
    mn, na, nb, nc, _ = MFMAS[name]
    return "%s %s, %s, %s, %s" % (mn, rng(d, nc), rng(a, na), rng(b, nb), rng(c, nc))


def probes_for(name):
    """(test, r) -> (first, second, copy-out) instruction texts."""
    mn, na, nb, nc, _ = MFMAS[name]
    m = mfma(name, D0, A0, B0, C0)
    out = {}
    copy_d = "\n".join("v_mov_b32 v%d, v%d" % (R0 + i, D0 + i) for i in range(nc))
    copy_e = "\n".join("v_mov_b32 v%d, v%d" % (R0 + i, E0 + i) for i in range(nc))
    for r in sorted({0, 1, nc - 1}):
        out[("war_c", r)] = (m, "v_mov_b32 v%d, v%d" % (C0 + r, JUNK), copy_d)                 # MFMA reads C, vector write of C
        out[("raw_d", r)] = (m, "v_mov_b32 v%d, v%d" % (TMP, D0 + r), "v_mov_b32 v%d, v%d" % (R0, TMP))   # vector read of D
        out[("waw_d", r)] = (m, "v_mov_b32 v%d, v%d" % (D0 + r, JUNK), copy_d)                 # vector write of D
        out[("vw_c", r)] = ("v_mov_b32 v%d, v%d" % (C0 + r, B0), m, copy_d)                    # vector write -> MFMA reads it as C
    for r in sorted({0, 1, na - 1}):
        out[("war_a", r)] = (m, "v_mov_b32 v%d, v%d" % (A0 + r, JUNK), copy_d)                 # MFMA reads A, vector write of A
        out[("vw_a", r)] = ("v_mov_b32 v%d, v%d" % (A0 + r, B0 + r), m, copy_d)                # vector write -> MFMA reads it as A
    # dependent chains: does a queued MFMA read its operands when it is issued, or when its predecessor has finished?
    #   ch_c: M1 -> D; M2 = A B + D -> E; vector write of D[r]          acc_aN / acc_bN: N MFMAs accumulating into D, the last one with
    #   ch_a: the same, M2's A operand in other registers, written       its own A (B) registers, which a vector instruction then writes
    a2 = 90 if na <= 4 else None
    if a2 is not None:
        seta = "\n".join("v_mov_b32 v%d, v%d" % (a2 + i, B0 + i) for i in range(na))
        m2e = "%s\n%s %s, %s, %s, %s" % (m, mn, rng(E0, nc), rng(A0, na), rng(B0, nb), rng(D0, nc))
        m2a = "%s\n%s %s, %s, %s, %s" % (m, mn, rng(E0, nc), rng(a2, na), rng(B0, nb), rng(D0, nc))
        for r in sorted({0, nc - 1}):
            out[("ch_c", r)] = (seta + "\ns_nop 15\n" + m2e, "v_mov_b32 v%d, v%d" % (D0 + r, JUNK), copy_e)
        for r in sorted({0, na - 1}):
            out[("ch_a", r)] = (seta + "\ns_nop 15\n" + m2a, "v_mov_b32 v%d, v%d" % (a2 + r, JUNK), copy_e)
            for n in (2, 3, 4):
                acc = "\n".join([m] + ["%s %s, %s, %s, %s" % (mn, rng(D0, nc), rng(A0, na), rng(B0, nb), rng(D0, nc))] * (n - 2))
                lasta = "%s %s, %s, %s, %s" % (mn, rng(D0, nc), rng(a2, na), rng(B0, nb), rng(D0, nc))
                lastb = "%s %s, %s, %s, %s" % (mn, rng(D0, nc), rng(A0, na), rng(a2, nb), rng(D0, nc))
                out[("acc_a%d" % n, r)] = (seta + "\ns_nop 15\n" + acc + "\n" + lasta, "v_mov_b32 v%d, v%d" % (a2 + r, JUNK), copy_d)
                out[("acc_b%d" % n, r)] = (seta + "\ns_nop 15\n" + acc + "\n" + lastb, "v_mov_b32 v%d, v%d" % (a2 + r, JUNK), copy_d)
        # the same with d wait states between the two MFMAs (0: the first one's result can be forwarded; more: the second one
        # has to wait for it in the register file) and, `m`, a third MFMA accumulating into E behind the second
        m2only = "%s %s, %s, %s, %s" % (mn, rng(E0, nc), rng(A0, na), rng(B0, nb), rng(D0, nc))
        m3 = "%s %s, %s, %s, %s" % (mn, rng(E0, nc), rng(A0, na), rng(B0, nb), rng(E0, nc))
        for d in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12):
            for r in sorted({0, 1, nc - 1}):
                gap = "\n".join(["s_nop 0"] * d)
                out[("cd%02d_c" % d, r)] = (m + "\n" + gap + "\n" + m2only, "v_mov_b32 v%d, v%d" % (D0 + r, JUNK), copy_e)
                out[("cd%02dm_c" % d, r)] = (m + "\n" + gap + "\n" + m2only + "\n" + m3, "v_mov_b32 v%d, v%d" % (D0 + r, JUNK), copy_e)
    # accumulate chain with a gap: M1 -> D; k wait states; M2 accumulates into D (same shape / the k = 16 shape of the family)
    out[("acg_same", 0)] = (m, mfma(name, D0, A0, B0, D0), copy_d)
    half = {"bf16_16x16x32": "bf16_16x16x16", "f16_16x16x32": "f16_16x16x16"}.get(name)
    if half:
        # absolute reference for the chain: the two products separately (the second one on a zero accumulator, written to E)
        out[("ref_m1", 0)] = (m, "s_nop 0", copy_d)
        out[("ref_m2", 0)] = ("s_nop 0", mfma(half, E0, A0, B0, E0), copy_e)
        pre_d = "\n".join("v_mov_b32 v%d, v%d" % (D0 + i, C0 + i) for i in range(nc)) + "\ns_nop 15\n"
        out[("acg_h_cd", 0)] = (pre_d + mfma(name, D0, A0, B0, D0), mfma(half, D0, A0, B0, D0), copy_d)          # first MFMA accumulates too (C = D)
        out[("acg_h_c0", 0)] = ("%s %s, %s, %s, 0" % (mn, rng(D0, nc), rng(A0, na), rng(B0, nb)), mfma(half, D0, A0, B0, D0), copy_d)   # first MFMA: C = 0
        out[("acg_h_sep", 0)] = (m, mfma(half, D0, A0 + 4, B0 + 4, D0), copy_d)                                  # second MFMA: other A / B registers
        out[("acg_h_e", 0)] = (m, mfma(half, E0, A0, B0, D0), copy_e)                                            # second MFMA: other destination
        out[("acg_rev", 0)] = (mfma(half, D0, A0, B0, C0), mfma(name, D0, A0, B0, D0), copy_d)                   # k = 16 first, k = 32 accumulates
        out[("acg_half", 0)] = (m, mfma(half, D0, A0, B0, D0), copy_d)
        # the kernel's shape: M1 -> D (k = 32); 7 states; M2 = A B + D -> E (k = 32); FILL; M3 accumulates into E (k = 16)
        out[("acg_k3", 0)] = (m + "\n" + "\n".join(["s_nop 0"] * 7) + "\n" + mfma(name, E0, A0, B0, D0), mfma(half, E0, A0, B0, E0), copy_e)
    # round 6: a two-step ACCUMULATING chain (M1 -> D; k1 vector fillers; M2 = A B + D -> D), then a vector read of D[r]: is the distance
    # hipcc keeps behind M2 (the plain raw_d figure) enough when M2 itself has to wait for M1 (inside the pipe, or queued behind another wave's)?
    for k1 in (0, 1, 2, 4, 7):
        for r in sorted({0, 1, nc - 1}):
            gap = "\n".join(["v_mov_b32 v%d, v%d" % (F_V, F_VS)] * k1)
            out[("chr%d" % k1, r)] = (m + "\n" + gap + "\n" + mfma(name, D0, A0, B0, D0), "v_mov_b32 v%d, v%d" % (TMP, D0 + r), "v_mov_b32 v%d, v%d" % (R0, TMP))
    # round 6: an LDS read that RETURNS INTO registers of an MFMA issued shortly before it (profiles/r06_bwd.md): the data comes back
    # some 64+ cycles after the request, which is what hipcc (4 states for a late C read, nothing for D) and round 5's table (vector
    # writes only) rely on — is that enough when the MFMA has to wait for a predecessor, or for the matrix pipe?
    #   ldc: M reads C, ds_read into C[r]            ldcd: M1 -> D; M2 = A B + D -> E (has to wait for M1); ds_read into D[r] (M2's C)
    #   ldd: M writes D, ds_read into D[r] (the LDS data must survive)      lddd: M1 -> D; M2 accumulates into D; ds_read into D[r]
    for r in sorted({0, nc - 1}):
        ld = "ds_read_b32 v%d, v%d"
        out[("ldc", r)] = (m, ld % (C0 + r, F_DSA), copy_d)
        out[("ldcd", r)] = (m + "\n" + mfma(name, E0, A0, B0, D0), ld % (D0 + r, F_DSA), copy_e)
        out[("ldd", r)] = (m, ld % (D0 + r, F_DSA), copy_d)
        out[("lddd", r)] = (m + "\n" + mfma(name, D0, A0, B0, D0), ld % (D0 + r, F_DSA), copy_d)
    out[("xd_a", 0)] = (m, mfma(name, E0, D0, B0, C0), copy_e)                                 # MFMA D -> next MFMA's A
    out[("xd_c", 0)] = (m, mfma(name, E0, A0, B0, D0), copy_e)                                 # MFMA D -> next MFMA's C, other destination
    return out


def perm_probes():
    out = {}
    setup = "v_mov_b32 v90, v%d" % A0   # (v90.. double as the chain probes' second operand block)
    for w in (32, 16):
        out[("perm%d" % w, 0)] = ("v_mov_b32 v91, v%d" % B0, "v_permlane%d_swap_b32 v90, v91" % w,
                                  "v_mov_b32 v%d, v90\nv_mov_b32 v%d, v91" % (R0, R0 + 1))
        out[("perm%d_dst" % w, 0)] = ("v_mov_b32 v90, v%d" % (B0 + 1), "v_permlane%d_swap_b32 v90, v91" % w,
                                      "v_mov_b32 v%d, v90\nv_mov_b32 v%d, v91" % (R0, R0 + 1))
        # round 6: the swap's RESULT read by the next vector instruction / by the other swap (csrc/sta_xattn_proj3.hip::bcast_row2 pads both)
        out[("perm%d_rd" % w, 0)] = ("v_mov_b32 v91, v%d\ns_nop 4\nv_permlane%d_swap_b32 v90, v91" % (B0, w), "v_mov_b32 v%d, v91" % TMP,
                                     "v_mov_b32 v%d, v%d\nv_mov_b32 v%d, v90" % (R0, TMP, R0 + 1))
        out[("perm%d_sw" % w, 0)] = ("v_mov_b32 v91, v%d\ns_nop 4\nv_permlane%d_swap_b32 v90, v91" % (B0, w), "v_permlane%d_swap_b32 v91, v90" % (48 - w),
                                     "v_mov_b32 v%d, v90\nv_mov_b32 v%d, v91" % (R0, R0 + 1))
    return setup, out


AGG0 = 130      # accumulators of the aggressor waves


def aggressor(name):
    """Waves 4..7 of the workgroup (the second wave of every SIMD) keep the matrix pipe busy with independent MFMAs while waves
    0..3 run the probe: does an MFMA that has to queue behind another wave's read its operands later than one that does not?"""
    mn, na, nb, nc, _ = MFMAS[name]
    blocks = 8 if nc == 4 else 4
    ms = ["%s %s, %s, %s, %s" % (mn, rng(AGG0 + nc * i, nc), rng(A0, na), rng(B0, nb), rng(AGG0 + nc * i, nc)) for i in range(blocks)]
    return ["s_cmp_lt_u32 %4, 4", "s_cbranch_scc1 L_victim_%=", "s_mov_b32 s41, 120", "L_agg_%=:"] + ms + \
           ["s_sub_u32 s41, s41, 1", "s_cmp_lg_u32 s41, 0", "s_cbranch_scc1 L_agg_%=", "s_branch L_end_%=", "L_victim_%=:"] + ["s_nop 15"] * 16


def kernel_text(idx, pre, first, fill, second, copy, contend=None):
    body = ["global_load_dwordx4 v[%d:%d], %%0, %%1" % (A0, A0 + 3), "global_load_dwordx4 v[%d:%d], %%0, %%1 offset:1024" % (A0 + 4, A0 + 7),
            "global_load_dwordx4 v[%d:%d], %%0, %%1 offset:2048" % (B0, B0 + 3), "global_load_dwordx4 v[%d:%d], %%0, %%1 offset:3072" % (B0 + 4, B0 + 7)]
    for i in range(4):
        body.append("global_load_dwordx4 v[%d:%d], %%0, %%2 offset:%d" % (C0 + 4 * i, C0 + 4 * i + 3, 1024 * i))
    body += ["v_mov_b32 v%d, 0x4a4a4a4a" % JUNK, "v_mov_b32 v%d, 0" % F_VS, "v_lshrrev_b32 v%d, 2, %%0" % F_DSA, "v_mov_b32 v91, 0"]
    body += ["v_mov_b32 v%d, 0" % (R0 + i) for i in range(16)]
    body += ["v_mov_b32 v%d, 0" % (D0 + i) for i in range(16)] + ["v_mov_b32 v%d, 0" % (E0 + i) for i in range(16)]
    body += ["s_waitcnt vmcnt(0)", "s_nop 15"]
    if contend:
        body += ["v_mov_b32 v%d, 0" % (AGG0 + i) for i in range(64)] + ["s_nop 4"] + aggressor(contend)
    if pre:
        body += [pre, "s_nop 15"]
    body += [first, fill, second] + ["s_nop 15"] * (8 if contend else 2) + ["s_waitcnt lgkmcnt(0)", copy, "s_nop 4"]
    for i in range(4):
        body.append("global_store_dwordx4 %%0, v[%d:%d], %%3 offset:%d" % (R0 + 4 * i, R0 + 4 * i + 3, 1024 * i))
    body += ["s_waitcnt vmcnt(0)"] + (["L_end_%=:"] if contend else [])
    text = "\\n\\t".join(x for b in body for x in b.split("\n") if x)
    clob = ", ".join('"v%d"' % i for i in range(A0, AGG0 + 72)) + ', "s40", "s41", "memory"'
    return ('extern "C" __global__ __launch_bounds__(512) void k%d(const char* in, char* out) {\n'
            '  __shared__ volatile unsigned lds[64]; if (threadIdx.x < 64) lds[threadIdx.x] = threadIdx.x; __syncthreads();\n'
            '  const unsigned off = (threadIdx.x & 63) * 16;\n'
            '  char* o = out + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 4096;\n'
            '  asm volatile("%s" :: "v"(off), "s"(in), "s"(in + 4096), "s"(o), "s"(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) : %s);\n}\n' % (idx, text, clob))


def generate(kmax4, kmax8, contend=False, only=None):
    variants = []      # (mfma, test, r, filler, k)  k = -1: the reference
    texts = []

    def add(key, pre, first, fill, second, copy):
        variants.append(key)
        texts.append(kernel_text(len(variants) - 1, pre, first, fill, second, copy, contend=(key[0] if contend and key[0] in MFMAS else None)))

    for name in MFMAS:
        kmax = kmax8 if MFMAS[name][4] == 8 else kmax4
        for (test, r), (first, second, copy) in probes_for(name).items():
            if only and not re.search(only, test):
                continue
            add((name, test, r, "gold", -1), None, first, GOLD, second, copy)
            for fn, ft in FILLERS.items():
                for k in range(kmax + 1):
                    add((name, test, r, fn, k), None, first, "\n".join([ft] * k), second, copy)
    setup, pp = perm_probes()
    for (test, r), (first, second, copy) in pp.items():
        if only and not re.search(only, test):
            continue
        add(("-", test, r, "gold", -1), setup, first, GOLD, second, copy)
        for fn, ft in FILLERS.items():
            for k in range(6):
                add(("-", test, r, fn, k), setup, first, "\n".join([ft] * k), second, copy)
    src = "#include <hip/hip_runtime.h>\n" + "".join(texts)
    return variants, src


def make_input(seed=3):
    """8 blocks of 1 KiB (16 bytes per lane): A (2 blocks: 16-bit pairs, bytes double as fp8), B (2), C (4: floats)."""
    rnd = random.Random(seed)
    buf = bytearray()
    for blk in range(4):
        for _ in range(64 * 8):
            # 16-bit patterns that are moderate numbers as fp16 (1 .. 3) and as bf16 (0.008 .. 4) and whose bytes are finite e4m3
            h = (rnd.getrandbits(1) << 15) | (rnd.randrange(0x3c, 0x41) << 8) | rnd.getrandbits(8)
            if (h & 0x7f) == 0x7f:
                h ^= 0x01                      # no e4m3 NaN in the low byte (the high byte never is one)
            buf += struct.pack("<H", h)
    for blk in range(4):
        for _ in range(64 * 4):
            buf += struct.pack("<f", rnd.uniform(-4.0, 4.0))
    return bytes(buf)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/hazard_probe.txt")
    ap.add_argument("--workdir", default="/tmp/hazard_probe")
    ap.add_argument("--kmax4", type=int, default=10)
    ap.add_argument("--kmax8", type=int, default=14)
    ap.add_argument("--compile-only", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--detail", action="store_true", help="per failing probe: which result registers / lane rows differ, and by how much")
    ap.add_argument("--only", default=None, help="regex over probe names (e.g. '^cd')")
    ap.add_argument("--contend", action="store_true", help="waves 4..7 saturate the matrix pipe while waves 0..3 run the probe")
    a = ap.parse_args()
    os.makedirs(a.workdir, exist_ok=True)
    variants, src = generate(a.kmax4, a.kmax8, a.contend, a.only)
    nfiles = 16
    per = (len(variants) + nfiles - 1) // nfiles
    chunks = src.split('extern "C"')
    head, kernels = chunks[0], ['extern "C"' + c for c in chunks[1:]]
    procs = []
    for f in range(nfiles):
        path = os.path.join(a.workdir, "hp%d.hip" % f)
        with open(path, "w") as fh:
            fh.write(head + "".join(kernels[f * per:(f + 1) * per]))
        co = os.path.join(a.workdir, "hp%d.co" % f)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "--cuda-device-only", "--no-gpu-bundle-output", "-o", co, path]))
    for p in procs:
        if p.wait() != 0:
            sys.exit("hipcc failed")
    print("%d probes compiled" % len(variants), flush=True)
    if a.compile_only:
        return
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")

    def ck(e, what):
        if e != 0:
            sys.exit("%s failed: %d" % (what, e))

    inp = make_input()
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    ck(hip.hipMalloc(ctypes.byref(d_in), len(inp)), "hipMalloc")
    nwaves = 8
    obytes = nwaves * 4096
    ck(hip.hipMalloc(ctypes.byref(d_out), obytes), "hipMalloc")
    ck(hip.hipMemcpy(d_in, inp, len(inp), 1), "hipMemcpy")
    results = {}
    host = (ctypes.c_char * obytes)()
    for f in range(nfiles):
        mod = ctypes.c_void_p()
        with open(os.path.join(a.workdir, "hp%d.co" % f), "rb") as fh:
            image = fh.read()
        ck(hip.hipModuleLoadData(ctypes.byref(mod), image), "hipModuleLoadData")
        for idx in range(f * per, min((f + 1) * per, len(variants))):
            fn = ctypes.c_void_p()
            ck(hip.hipModuleGetFunction(ctypes.byref(fn), mod, b"k%d" % idx), "hipModuleGetFunction")
            outs = []
            for rep in range(a.reps):
                ck(hip.hipMemset(d_out, 0xee, obytes), "hipMemset")
                args = (ctypes.c_void_p * 2)(ctypes.cast(ctypes.pointer(d_in), ctypes.c_void_p), ctypes.cast(ctypes.pointer(d_out), ctypes.c_void_p))
                ck(hip.hipModuleLaunchKernel(fn, 1, 1, 1, 64 * nwaves, 1, 1, 0, None, args, None), "launch k%d" % idx)
                ck(hip.hipDeviceSynchronize(), "sync k%d" % idx)
                ck(hip.hipMemcpy(host, d_out, obytes, 2), "hipMemcpy")
                outs.append(bytes(host))
            results[variants[idx]] = outs
        hip.hipModuleUnload(mod)
    # every wave and repetition of the reference must agree with each other; a probe passes if all its waves equal the reference
    lines = []
    table = {}
    for key, outs in results.items():
        name, test, r, fn, k = key
        if fn != "gold":
            continue
        ref = outs[0][:4096]
        stable = all(o[w * 4096:(w + 1) * 4096] == ref for o in outs for w in range(4 if a.contend else nwaves))
        if not stable:
            lines.append("UNSTABLE reference %s %s r=%d" % (name, test, r))
        table[(name, test, r)] = ref
    summary = {}
    details = []
    for key, outs in sorted(results.items()):
        name, test, r, fn, k = key
        if fn == "gold":
            continue
        ref = table[(name, test, r)]
        ok = all(o[w * 4096:(w + 1) * 4096] == ref for o in outs for w in range(4 if a.contend else nwaves))
        summary.setdefault((name, test, r, fn), {})[k] = ok
        if not ok and a.detail:
            # which (result register, lane row) differ in wave 0 of the first repetition, and how far off the values are
            o = outs[0][:4096]
            cells = {}
            for i in range(4):
                for lane in range(64):
                    for j in range(4):
                        off = i * 1024 + lane * 16 + j * 4
                        if o[off:off + 4] != ref[off:off + 4]:
                            x, y = struct.unpack("<f", o[off:off + 4])[0], struct.unpack("<f", ref[off:off + 4])[0]
                            c = cells.setdefault((4 * i + j, lane >> 4), [0, 0.0])
                            c[0] += 1
                            c[1] = max(c[1], abs(x - y) / (abs(y) + 1e-30))
            details.append("%-14s %-10s r=%-2d %-5s k=%-2d " % (name, test, r, fn, k) + " ".join("R%d/row%d:%d(%.1e)" % (rg, row, n, e) for (rg, row), (n, e) in sorted(cells.items())))
    lines.append("# gfx950 hazard probe: per (MFMA, pair, register, filler): pass map over k = 0.. wait states, and the smallest k from which every probe passes")
    lines.append("# pairs: war_c = MFMA reads C, vector write of C[r]; raw_d = vector read of D[r]; waw_d = vector write of D[r]; war_a = vector write of A[r];")
    lines.append("#        vw_a / vw_c = vector write of A[r] / C[r], then the MFMA; xd_a / xd_c = MFMA D -> the next MFMA's A / C (other destination); perm* = vector write -> v_permlane*_swap source (_dst: destination)")
    need = {}
    for (name, test, r, fn), m in sorted(summary.items()):
        ks = sorted(m)
        bits = "".join("." if m[k] else "X" for k in ks)
        first_ok = next((k for k in ks if all(m[j] for j in ks if j >= k)), None)
        lines.append("%-14s %-10s r=%-2d %-5s %s  need %s" % (name, test, r, fn, bits, first_ok if first_ok is not None else ">%d" % ks[-1]))
        cur = need.get((name, test, fn), 0)
        need[(name, test, fn)] = max(cur, first_ok if first_ok is not None else ks[-1] + 1)
    if details:
        lines.append("")
        lines.append("# failing probes: result register / lane row : lanes that differ (largest relative difference)")
        lines += details
    # absolute check of the mixed-shape accumulate chain: which of {short gap, long gap} equals product 1 + product 2 ?
    for name in MFMAS:
        if (name, "ref_m1", 0) in table and (name, "acg_half", 0) in table:
            f = lambda b: struct.unpack("<1024f", b)
            d1, d2, gold = f(table[(name, "ref_m1", 0)]), f(table[(name, "ref_m2", 0)]), f(table[(name, "acg_half", 0)])
            lines.append("")
            lines.append("# %s then its k = 16 shape accumulating into the same registers, against (product 1 + product 2): largest |difference| per result register" % name)
            for tag, got in [("32 wait states between", gold)] + [("k = %d (nop)" % k, f(results[(name, "acg_half", 0, "nop", k)][0][:4096])) for k in (0, 2, 4, 5, 6)]:
                errs = []
                for rg in range(4):
                    idx = [lane * 4 + rg for lane in range(64)]          # store 0 holds registers 0..3: dword lane * 4 + rg
                    errs.append(max(abs(got[i] - (d1[i] + d2[i])) for i in idx))
                lines.append("  %-24s %s" % (tag, "  ".join("R%d %.2e" % (rg, e) for rg, e in enumerate(errs))))
    lines.append("")
    lines.append("# wait states needed (max over registers), by filler")
    for (name, test, fn), v in sorted(need.items()):
        lines.append("%-14s %-10s %-5s %d" % (name, test, fn, v))
    text = "\n".join(lines) + "\n"
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as fh:
        fh.write(text)
    print(text[-6000:])


if __name__ == "__main__":
    main()
