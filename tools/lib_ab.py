"""A/B harness for kernel experiments: builds the library once per flag set into gpurun_out/libsta_<tag>.so and runs a
tool script against each build in its own process, `rounds` times interleaved.
usage: lib_ab.py <tool.py> [tool args ...] -- tag1=-DFLAG1,-DFLAG2 tag2= ...   (an empty flag list = the product build;
"src:<file>=<path>" in a flag list swaps one product source for an experiment build, see below)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import lib  # noqa: E402

sep = sys.argv.index("--")
tool, variants = sys.argv[1:sep], sys.argv[sep + 1:]
rounds = int(os.environ.get("AB_ROUNDS", "2"))
libs = {}
for v in variants:
    tag, flags = v.split("=", 1)
    out = os.path.join(ROOT, "build", "ab", "libsta_%s.so" % tag)      # build/ is git-ignored but travels to the GPU box
    os.makedirs(os.path.dirname(out), exist_ok=True)
    libs[tag] = out
    if os.path.exists(out) and os.environ.get("AB_REBUILD") != "1":
        continue
    objs = []
    flag_list = [f for f in flags.split(",") if f]
    # "src:<product file>=<path>" swaps one translation unit for an experiment build of it (a modified copy kept outside csrc/), e.g.
    #   try=src:sta_xattn_proj3.hip=/tmp/sta_xattn_proj3_try.hip
    swaps = dict(f[4:].split("=", 1) for f in flag_list if f.startswith("src:"))
    flag_list = [f for f in flag_list if not f.startswith("src:")]
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    work = tempfile.mkdtemp(prefix="sta_ab_")
    base = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", lib.INCLUDE, "-I", lib.CSRC]
    jobs = []
    with ThreadPoolExecutor(max_workers=len(lib.SOURCES)) as ex:
        for src in lib.SOURCES:
            obj = out + "." + os.path.basename(src) + ".o"
            b = os.path.basename(src)
            if b in swaps:
                src = os.path.join(ROOT, swaps[b])
            # every variant goes through the product build's pipeline: assembly -> hazard lint + s_nop cure (sta/isa_lint.py) -> assemble
            w = os.path.join(work, b)
            os.makedirs(w, exist_ok=True)
            jobs.append((ex.submit(lib._compile_checked, "hipcc", base, lib.PER_SOURCE_FLAGS.get(b, []) + flag_list, src, obj, w, False), obj))
        for fut, _ in jobs:
            fut.result()
    objs = [(None, o) for _, o in jobs]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *[o for _, o in objs], "-o", out])
    for _, o in objs:
        os.remove(o)
    libs[tag] = out
for r in range(rounds):
    for tag, path in libs.items():
        code = "import sys; sys.argv=%r; sys.path.insert(0, %r); from sta import lib; lib.LIB_PATH=%r; __file__=%r; exec(open(__file__).read())" % (
            tool, os.path.join(ROOT, "diffusion-spacetime-attn_amd"), path, os.path.abspath(tool[0]))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        for line in out.stdout.strip().splitlines():
            print("[%s r%d] %s" % (tag, r, line))
        if out.returncode:
            print(out.stderr[-500:])
