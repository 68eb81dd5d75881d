"""Bisect a miscompiled kernel at the assembly level: build the library with ONE source going through
    hipcc -S (device)  ->  a text patch of chosen kernels  ->  assembler  ->  lld  ->  bundle  ->  host compile with that bundle
once per patch, then run a tool (default tools/dbg_pair.py) against every build.

  build (no GPU needed):  python tools/asm_patch_ab.py build  [--source sta_xattn_proj3.hip] [--kernel REGEX] tag=patch[,patch] ...
  run   (GPU box):        python tools/asm_patch_ab.py run [tool.py args ...]

patches (applied to the instruction stream of the kernels whose mangled name matches --kernel):
  none                 the compiler's text, through the same pipeline (control)
  store_wait           s_waitcnt vmcnt(0) behind every buffer_store / global_store
  store_nop            s_nop 7 behind every buffer_store / global_store
  mfma_nop             s_nop 7 behind every v_mfma
  mfma_pre             s_nop 1 in front of every v_mfma
  asm_pad              s_nop 7 in front of and behind every asm statement
  lgkm0                every s_waitcnt lgkmcnt(N) / vmcnt(N) becomes a full wait
  cvt_nop              s_nop 1 behind every v_cvt_pk_bf16_f32
  perm_nop             s_nop 7 in front of and behind every v_permlane*_swap
  before:PREFIX:TEXT / after:PREFIX:TEXT   TEXT in front of / behind every instruction that starts with PREFIX
  range:A:B:TEXT       TEXT behind every instruction of kernel lines A..B (1-based inside the kernel)
  line:N:TEXT          insert TEXT in front of line N of the kernel's text (1-based inside the kernel)
  flag:-DNAME=V        not a text patch: an extra hipcc flag for this build of the source (ablation switches); the text then passes
                       the library's own lint + cure (sta/isa_lint.py) like a product build
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import lib  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"
OUT = os.path.join(ROOT, "build", "asm")


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def patch_kernel(lines, patches):
    out = []
    recent = []          # (kernel line, role -> registers) of the MFMAs seen so far
    for n, l in enumerate(lines, 1):
        s = l.strip()
        pre, post = [], []
        if s.startswith("v_mfma"):
            ops = [t.strip() for t in s.split(None, 1)[1].split(",")]
            recent.append((n, {"D": _regs(ops[0]), "A": _regs(ops[1]), "B": _regs(ops[2]), "C": _regs(ops[3])}))
        for p in patches:
            # dswar:ROLES:N:TEXT  TEXT in front of every ds_read / buffer_load whose destination overlaps a register that one of the
            # MFMAs of the previous N lines uses in one of ROLES (e.g. ABCD, D, AB)
            if p.startswith("dswar:") and s.startswith(("ds_read", "buffer_load", "global_load")):
                _, roles, dist, text = p.split(":", 3)
                dst = _regs(s.split(None, 1)[1].split(",")[0].strip())
                if any(n - ln <= int(dist) and any(dst & r[k] for k in roles) for ln, r in recent):
                    pre += ["\t" + t for t in text.split(";")]
        for p in patches:
            if p == "none":
                continue
            if p == "store_wait" and s.startswith(("buffer_store", "global_store")):
                post.append("\ts_waitcnt vmcnt(0)")
            elif p == "store_nop" and s.startswith(("buffer_store", "global_store")):
                post.append("\ts_nop 7")
            elif p == "mfma_nop" and s.startswith("v_mfma"):
                post.append("\ts_nop 7")
            elif p == "mfma_pre" and s.startswith("v_mfma"):
                pre.append("\ts_nop 1")
            elif p == "asm_pad" and s.startswith(";;#ASMSTART"):
                pre.append("\ts_nop 7")
            elif p == "asm_pad" and s.startswith(";;#ASMEND"):
                post.append("\ts_nop 7")
            elif p == "lgkm0" and s.startswith("s_waitcnt"):
                l = "\ts_waitcnt vmcnt(0) lgkmcnt(0)"
            elif p == "cvt_nop" and s.startswith("v_cvt_pk_bf16_f32"):
                post.append("\ts_nop 1")
            elif p == "perm_nop" and s.startswith("v_permlane"):
                pre.append("\ts_nop 7")
                post.append("\ts_nop 7")
            elif p.startswith(("before:", "after:")):      # before:PREFIX:TEXT / after:PREFIX:TEXT around every instruction that starts with PREFIX
                kind, prefix, text = p.split(":", 2)
                if s.startswith(prefix):
                    (pre if kind == "before" else post).append("\t" + text)
            elif p.startswith("range:"):          # range:A:B:TEXT  TEXT behind every instruction of kernel lines A..B
                _, a, b, text = p.split(":", 3)
                if int(a) <= n <= int(b) and s and not s.startswith((";", ".", "//")) and not s.endswith(":"):
                    post.append("\t" + text.replace("\\n", "\n\t"))
            elif p.startswith("line:"):
                _, ln, text = p.split(":", 2)
                if int(ln) == n:
                    pre.append("\t" + text.replace("\\n", "\n\t"))
        out += pre + [l] + post
    return out


def patch_text(text, kernel_re, patches):
    lines = text.split("\n")
    out, i = [], 0
    while i < len(lines):
        m = re.match(r"^(_Z[\w$.]*):", lines[i])
        if m and re.search(kernel_re, m.group(1)):
            j = i + 1
            while j < len(lines) and not lines[j].strip().startswith("s_endpgm"):
                j += 1
            out.append(lines[i])
            out += patch_kernel(lines[i + 1:j], patches)
            i = j
        else:
            out.append(lines[i])
            i += 1
    return "\n".join(out)


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit("failed: %s\n%s" % (" ".join(cmd), r.stdout + r.stderr))


def build(argv):
    source, kernel_re = "sta_xattn_proj3.hip", "IDF16bLi10ELi2ELb1"
    variants = []
    it = iter(argv)
    for a in it:
        if a == "--source":
            source = next(it)
        elif a == "--kernel":
            kernel_re = next(it)
        else:
            variants.append(a)
    os.makedirs(os.path.join(OUT, "common"), exist_ok=True)
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", lib.INCLUDE, "-I", lib.CSRC]
    procs, common = [], []
    for src in lib.SOURCES:
        b = os.path.basename(src)
        if b == source:
            continue
        obj = os.path.join(OUT, "common", b + ".o")
        common.append(obj)
        if not os.path.exists(obj) or os.path.getmtime(obj) < os.path.getmtime(src):
            procs.append(subprocess.Popen(base + lib.PER_SOURCE_FLAGS.get(b, []) + ["-c", src, "-o", obj]))
    src = os.path.join(lib.CSRC, source)
    flags = lib.PER_SOURCE_FLAGS.get(source, [])
    dev_s = os.path.join(OUT, source + ".s")
    run(base + flags + ["--cuda-device-only", "-S", src, "-o", dev_s])
    text = open(dev_s).read()
    for p in procs:
        assert p.wait() == 0
    from sta import isa_lint
    for v in variants:
        tag, plist = v.split("=", 1)
        w = os.path.join(OUT, tag)
        os.makedirs(w, exist_ok=True)
        patched = os.path.join(w, "dev.s")
        plist = [p for p in plist.split(",") if p]
        extra = [p[5:] for p in plist if p.startswith("flag:")]
        vtext = text
        if extra:
            run(base + flags + extra + ["--cuda-device-only", "-S", src, "-o", patched + ".raw"])
            vtext = isa_lint.fix_text(open(patched + ".raw").read())[0]
        with open(patched, "w") as fh:
            fh.write(patch_text(vtext, kernel_re, [p for p in plist if not p.startswith("flag:")]))
        run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", patched, "-o", w + "/dev.o"])
        run([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", w + "/dev.o", "-o", w + "/dev.out"])
        run([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
             "-input=/dev/null", "-input=" + w + "/dev.out", "-output=" + w + "/dev.hipfb"])
        run(base + flags + extra + ["--cuda-host-only", "-c", src, "-Xclang", "-fcuda-include-gpubinary", "-Xclang", w + "/dev.hipfb", "-o", w + "/host.o"])
        run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *common, w + "/host.o", "-o", os.path.join(OUT, "libsta_%s.so" % tag)])
        print("built", tag, flush=True)


def run_tool(argv):
    tool = argv or [os.path.join(ROOT, "tools", "dbg_pair.py")]
    for f in sorted(os.listdir(OUT)):
        if not (f.startswith("libsta_") and f.endswith(".so")):
            continue
        print("=== %s" % f[7:-3], flush=True)
        code = "import sys; sys.argv=%r; sys.path.insert(0, %r); from sta import lib; lib.LIB_PATH=%r; __file__=%r; exec(open(__file__).read())" % (
            tool, os.path.join(ROOT, "diffusion-spacetime-attn_amd"), os.path.join(OUT, f), tool[0])
        subprocess.run([sys.executable, "-c", code])


if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run_tool)(sys.argv[2:])
