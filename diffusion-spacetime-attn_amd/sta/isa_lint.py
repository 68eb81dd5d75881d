"""Offline hazard lint of the compiled gfx950 kernels (runs without a GPU: it reads hipcc's assembly).

hipcc pads the data hazards its recogniser models, but it does not look inside an `asm` statement (every instruction of the
statement counts as zero wait states, nothing in it is padded) and the level-0 kernels use such statements next to MFMAs.
This module re-checks every kernel of a source file against the hazard table MEASURED on MI355X by tools/hazard_probe.py
(profiles/r05_hazard_table.md): for each producer instruction it walks the control-flow graph forward and counts the wait
states (one per instruction, N + 1 per `s_nop N`) in front of every consumer inside the producer's window.

    pair                                                   wait states needed (measured; the lint's table = measured, rounded up to
                                                           what hipcc itself assumes where that is larger)
    MFMA writes D  -> vector / memory instruction reads D   16x16x16: 7   16x16x32: 8   32x32x16, 16x16x128 f8f6f4: 12
    MFMA writes D  -> another MFMA reads it as A or B       16x16x16: 5   16x16x32: 7   f8f6f4: 11        (lint: as the row above)
    MFMA writes D  -> vector instruction / load writes D    16x16x16: 4   16x16x32: 5   32x32x16, f8f6f4: 9
    MFMA reads  C  -> vector instruction writes C           4-pass shapes: 0 (all of C is read in the first pass);
                                                            32x32x16: 4 for registers 4.. (one pass per four registers)
    MFMA reads A/B -> vector instruction writes A/B         0, also behind a queue of dependent MFMAs and beside a second wave that
                                                            saturates the SIMD's matrix pipe (the wave does not issue past a queued MFMA)
    MFMA writes D  -> an MFMA of the SAME shape reads it as C            0 (interlocked / forwarded), at every distance
    MFMA writes D  -> an MFMA of ANOTHER shape reads it as C  5, unless a third MFMA was issued in between  (lint: 6)
        (v_mfma_f32_16x16x32 -> v_mfma_f32_16x16x16 or the reverse, same or other destination: the consumer reads registers 0 and 1 of
        the tile before the producer has written them.  hipcc 7.2 pads this pair with 0 wait states when the destination is the same:
        that is the "miscompiled bf16 out-fragment instantiation" of profiles/r04_level0.md — one v_xor between the k = 32 and the
        k = 16 MFMA of a PV chain — and NOT a write-after-read on C as that note guessed; tools/asm_patch_ab.py bisected it.)
    vector write   -> MFMA reads it as A / B / C            2 / 2 / 1                                     (lint: 2)
    vector write   -> v_permlane{16,32}_swap operand        1                                             (lint: 2, hipcc's figure)
    vector write   -> DPP source                            (not probed)                                  (lint: 2, hipcc's figure)
    v_rcp / v_exp / ... write -> other vector instruction reads        (not probed)                       (lint: 1, hipcc's figure)

Every kind of filler instruction counted as a wait state in the probes (s_nop, vector, LDS, scalar; LDS and vector fillers even
saved one), so the lint counts one state per instruction; `strict=True` does not count s_waitcnt (hipcc does).

API: lint_text(asm_text) -> list of Finding; lint_source(path, flags) compiles with `hipcc -S --cuda-device-only` first.
"""
import os
import re
import shutil
import subprocess
import tempfile
from collections import namedtuple

Finding = namedtuple("Finding", "kernel line rule producer consumer have need")

# MFMA mnemonic -> (passes, registers of D)
_MFMA_RE = re.compile(r"^v_mfma_(?:f32|i32|f64)_(\d+)x(\d+)x(\d+)")
_TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_", "v_rcp_iflag")
MAX_WINDOW = 13
# round 6 (profiles/r06_bwd_race.md): an optional rule — wait states between an MFMA and an LDS / vector-memory load whose DESTINATION
# overlaps the MFMA's C or D registers. It was one hypothesis for the eight-wave backward's non-reproducible bits; the probes
# (tools/hazard_probe.py ^ld) found no such hazard for the 16x16 shapes and 900+ such sites in kernels that have always been bit-stable,
# so it is OFF (0) in the build; tools can switch it on to count sites.
LOAD_RETURN_STATES = 0


def _mfma_info(mn):
    m = _MFMA_RE.match(mn)
    if not m:
        return None
    M, N, Kd = int(m.group(1)), int(m.group(2)), int(m.group(3))
    flops = M * N * Kd
    if "f8f6f4" in mn or "fp8" in mn or "bf8" in mn:
        passes = 8 if (M == 16 and Kd >= 128) or (M == 32 and Kd >= 64) else 4
    elif M == 32:
        passes = 8 if Kd >= 8 else 16
    elif M == 16:
        passes = 4 if Kd >= 8 else 8
    else:
        passes = 2
    del flops
    return passes


def _shape(mn):
    m = _MFMA_RE.match(mn)
    return m.groups() if m else None


def _regs(tok):
    """'v[4:7]' -> ('v', 4..7); 'v12' -> ('v', 12); AGPRs likewise with 'a'. Anything else -> None."""
    tok = tok.strip().lstrip("-").strip("|")
    if tok.startswith(("abs(", "neg(")):
        tok = tok[4:].rstrip(")")
    m = re.match(r"^(v|a)\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^(v|a)(\d+)$", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return None


class Ins:
    __slots__ = ("line", "mn", "ops", "defs", "uses", "states", "mfma", "text", "in_asm", "dpp", "a", "b", "c")

    def __init__(self, line, text, in_asm):
        self.line, self.text, self.in_asm = line, text, in_asm
        body = text.split("//")[0].split(";")[0].strip()
        parts = body.split(None, 1)
        self.mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        # operands: split on commas outside brackets; modifiers (offset:.., op_sel..) carry no registers we track
        ops, depth, cur = [], 0, ""
        for ch in rest:
            if ch == "[":
                depth += 1
            elif ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                ops.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            ops.append(cur.strip())
        # a trailing operand may carry modifiers after a space: "v1 offset:16", "v[0:1] quad_perm:[..]"
        self.dpp = any(k in rest for k in ("quad_perm", "row_shl", "row_shr", "row_ror", "row_bcast", "row_mirror", "row_half_mirror", "wave_shl", "wave_ror", "row_newbcast", "row_share")) or self.mn.endswith("_dpp")
        self.ops = [o.split()[0] if o.split() else o for o in ops]
        self.states = 1
        if self.mn == "s_nop":
            self.states = int(self.ops[0], 0) + 1
        self.mfma = _mfma_info(self.mn)
        regs = [_regs(o) for o in self.ops]
        self.defs, self.uses = set(), set()
        self.a = self.b = self.c = set()
        mn = self.mn
        if self.mfma is not None:
            self.defs = regs[0] or set()
            self.a, self.b, self.c = regs[1] or set(), regs[2] or set(), (regs[3] or set()) if len(regs) > 3 else set()
            self.uses = self.a | self.b | self.c
        elif mn.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap_b32")):
            for r in regs[:2]:
                if r:
                    self.defs |= r
                    self.uses |= r
        elif mn.startswith(("global_store", "buffer_store", "flat_store", "ds_write", "ds_store", "scratch_store", "global_atomic", "buffer_atomic", "ds_add", "ds_max", "ds_min")) and "rtn" not in mn:
            for r in regs:
                if r:
                    self.uses |= r
        elif mn.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")):
            for r in regs[1:]:
                if r:
                    self.uses |= r
            if regs and regs[0] and not mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                self.defs |= regs[0]
        elif mn.startswith(("s_", ";")):
            pass
        else:       # vector ALU, loads (first operand = destination), LDS-DMA loads have no register destination
            if regs and regs[0] is not None and not (mn.startswith(("buffer_load", "global_load")) and (" lds" in (" " + rest) or "_lds_" in mn)):
                self.defs |= regs[0]
                for r in regs[1:]:
                    if r:
                        self.uses |= r
            else:
                for r in regs:
                    if r:
                        self.uses |= r
            if mn.startswith(("v_fmac", "v_mac", "v_dot2c", "v_pk_fmac", "v_cndmask")) or "accumulate" in mn:
                self.uses |= self.defs        # destination is also read
            if mn.startswith("v_mov") and self.dpp:
                self.uses |= set()            # old value only matters with bound_ctrl:0 absent; irrelevant to the table

    @property
    def is_valu(self):
        return self.mn.startswith("v_") and self.mfma is None

    @property
    def is_trans(self):
        return self.mn.startswith(_TRANS)

    @property
    def is_load(self):
        """LDS / vector-memory instruction whose result comes back into vector registers some time after it was issued."""
        return bool(self.defs) and self.mn.startswith(("ds_read", "ds_load", "buffer_load", "global_load", "flat_load", "scratch_load"))

    @property
    def is_vmem_or_lds(self):
        return self.mn.startswith(("global_", "buffer_", "flat_", "ds_", "scratch_"))


def _parse_kernels(text):
    """-> {kernel: (instructions, labels{name: index}, succ{index: [indices]})}"""
    kernels = {}
    cur, name, in_asm = None, None, False
    lines = text.splitlines()
    for ln, raw in enumerate(lines, 1):
        s = raw.strip()
        if not s:
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
        if m and not raw[0].isspace():
            lab = m.group(1)
            if not lab.startswith((".L", "L_")) and not lab.startswith("."):
                name = lab
                cur = {"ins": [], "labels": {}}
                kernels[name] = cur
            elif cur is not None:
                cur["labels"][lab] = len(cur["ins"])
            continue
        if cur is None:
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):\s*(;.*)?$", s)       # labels inside asm statements (indented)
        if m:
            cur["labels"][m.group(1)] = len(cur["ins"])
            continue
        if s.startswith((";", ".", "//")):
            if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
                cur = None
            continue
        if s.startswith("s_endpgm"):
            cur["ins"].append(Ins(ln, s, in_asm))
            continue
        cur["ins"].append(Ins(ln, s, in_asm))
    return kernels


def _successors(ins, labels, i):
    x = ins[i]
    if x.mn == "s_endpgm" or x.mn.startswith("s_setpc"):
        return []
    if x.mn == "s_branch":
        t = labels.get(x.ops[0])
        return [t] if t is not None and t < len(ins) else []
    out = [i + 1] if i + 1 < len(ins) else []
    if x.mn.startswith("s_cbranch"):
        t = labels.get(x.ops[0])
        if t is not None and t < len(ins):
            out.append(t)
    return out


def _counts(x, strict):
    if strict and x.mn == "s_waitcnt":
        return 0
    return x.states


def lint_text(text, strict=False, only=None):
    findings = []
    for kname, k in _parse_kernels(text).items():
        if only and not re.search(only, kname):
            continue
        ins, labels = k["ins"], k["labels"]
        n = len(ins)
        for i, p in enumerate(ins):
            rules = []       # (rule, registers, need, consumer predicate)
            if p.mfma is not None:
                P = p.mfma
                four = P <= 4
                small = four and p.mn.startswith(("v_mfma_f32_16x16x16", "v_mfma_f32_32x32x8"))
                need_raw = (7 if small else 8) if four else P + 4
                need_waw = (4 if small else 5) if four else P + 1
                rules.append(("mfma D -> read", p.defs, need_raw, lambda c, R: bool(c.uses & R) and not (c.mfma is not None and not ((c.a | c.b) & R)), False))
                rules.append(("mfma D -> write", p.defs, need_waw, lambda c, R: c.mfma is None and bool(c.defs & R), False))
                if LOAD_RETURN_STATES:
                    rules.append(("mfma C / D -> a load returns into it", p.defs | p.c, LOAD_RETURN_STATES, lambda c, R: c.is_load and bool(c.defs & R), False))
                rules.append(("mfma D -> C of an mfma of another shape", p.defs, 6, lambda c, R, pm=_shape(p.mn): c.mfma is not None and bool(c.c & R) and _shape(c.mn) != pm, True))
                if not four:
                    late = {r for r in p.c if r not in p.defs and (r[1] - min(x[1] for x in p.c)) >= 4}
                    if late:
                        rules.append(("mfma reads C -> write", late, 4, lambda c, R: c.mfma is None and bool(c.defs & R), False))
            elif p.is_valu and p.defs:
                rules.append(("vector write -> mfma operand", p.defs, 2, lambda c, R: c.mfma is not None and bool(c.uses & R), False))
                rules.append(("vector write -> permlane swap", p.defs, 2, lambda c, R: c.mn.startswith(("v_permlane16_swap", "v_permlane32_swap")) and bool((c.uses | c.defs) & R), False))
                rules.append(("vector write -> dpp source", p.defs, 2, lambda c, R: c.dpp and bool(c.uses & R), False))
                if p.is_trans:
                    rules.append(("transcendental -> vector read", p.defs, 1, lambda c, R: c.is_valu and not c.is_trans and bool(c.uses & R), False))
            if not rules:
                continue
            maxneed = max(r[2] for r in rules)
            # forward walk: (index, wait states so far, live register sets per rule)
            stack = [(j, 0, tuple(frozenset(r[1]) for r in rules)) for j in _successors(ins, labels, i)]
            seen = {}
            while stack:
                j, ws, live = stack.pop()
                if ws >= maxneed or j >= n:
                    continue
                key = (j, live)
                if seen.get(key, 1 << 30) <= ws:
                    continue
                seen[key] = ws
                c = ins[j]
                newlive = []
                for (rule, _, need, pred, mfma_clears), R in zip(rules, live):
                    if R and ws < need and pred(c, R):
                        findings.append(Finding(kname, c.line, rule, "%d: %s" % (p.line, p.text.split("//")[0].strip()), c.text.split("//")[0].strip(), ws, need))
                    if mfma_clears and c.mfma is not None:
                        R = frozenset()          # a third MFMA between the two serialises them on the matrix pipe: enough
                    # a register rewritten by the consumer is no longer the producer's value
                    newlive.append(frozenset(R - c.defs) if c.defs and not rule.endswith("write") else R)
                ws2 = ws + _counts(c, strict)
                for t in _successors(ins, labels, j):
                    stack.append((t, ws2, tuple(newlive)))
    # one finding per (kernel, consumer line, rule, producer)
    uniq, out = set(), []
    for f in findings:
        key = (f.kernel, f.line, f.rule, f.producer)
        if key not in uniq:
            uniq.add(key)
            out.append(f)
    return out


def fix_text(text, strict=False, max_rounds=8):
    """Insert `s_nop` in front of every consumer the lint flags until the text is clean. -> (text, [(line, states inserted, rule)])"""
    log = []
    for _ in range(max_rounds):
        findings = lint_text(text, strict=strict)
        if not findings:
            return text, log
        need = {}
        for f in findings:
            d = f.need - f.have
            if d > need.get(f.line, (0, ""))[0]:
                need[f.line] = (d, f.rule)
        lines = text.split("\n")
        for ln in sorted(need, reverse=True):
            d, rule = need[ln]
            pad = []
            while d > 0:
                pad.append("\ts_nop %d\t; sta isa_lint: %s" % (min(d, 16) - 1, rule))
                d -= min(d, 16)
            lines[ln - 1:ln - 1] = pad
            log.append((ln, need[ln][0], rule))
        text = "\n".join(lines)
    raise RuntimeError("isa_lint.fix_text did not converge:\n" + format_findings(lint_text(text, strict=strict)))


def compile_to_asm(src, flags, include_dirs, workdir=None):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    d = workdir or tempfile.mkdtemp(prefix="sta_lint_")
    out = os.path.join(d, os.path.basename(src) + ".s")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S"]
    for inc in include_dirs:
        cmd += ["-I", inc]
    cmd += list(flags) + [src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed: %s\n%s" % (" ".join(cmd), r.stderr))
    return out


def lint_source(src, flags=(), include_dirs=(), strict=False, only=None):
    path = compile_to_asm(src, flags, include_dirs)
    with open(path) as fh:
        text = fh.read()
    shutil.rmtree(os.path.dirname(path), ignore_errors=True)
    return lint_text(text, strict=strict, only=only)


def format_findings(findings):
    return "\n".join("%s: line %d: %s: `%s` %d wait state(s) behind `%s`, needs %d" % (f.kernel[:70], f.line, f.rule, f.consumer, f.have, f.producer, f.need)
                     for f in findings)
