"""ctypes binding of libsta_xattn.so (C-ABI declared in include/sta_xattn.h).

The library is built in-tree by :func:`build` (hipcc, gfx950 only) and loaded lazily by
:func:`load`. There is no fallback of any kind: if the shared object is missing or a symbol cannot
be resolved, :class:`StaLibraryError` is raised — the product path must never run without the HIP
kernels.
"""
import ctypes
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)                      # diffusion-spacetime-attn_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)
CSRC = os.path.join(PKG_ROOT, "csrc")
INCLUDE = os.path.join(REPO_ROOT, "include")
LIB_PATH = os.path.join(CSRC, "libsta_xattn.so")
SOURCES = [os.path.join(CSRC, n) for n in ("sta_xattn.hip", "sta_xattn_bwd.hip", "sta_xattn_proj.hip", "sta_xattn_proj3.hip", "sta_rowgemm.hip", "sta_ffgemm.hip", "sta_conv.hip", "sta_gemm.hip", "sta_selfattn.hip", "sta_selfattn_bwd.hip", "sta_unet.hip", "sta_unet_bwd.hip", "sta_fp8.hip")]

# Self-attention keeps its MFMA accumulators in VGPRs: hipcc otherwise parks them in AGPRs and brackets the
# online-softmax rescale with v_accvgpr_read/write pairs (120 extra VALU instructions per key block in a kernel
# that is VALU-issue bound: 859 -> 765 us at B=16, N=4096, d=40). The cross-attention kernels measured neutral
# (forward) to slower (backward at d=160, 26 -> 30 us) with it, so the flag is per source.
# The cross-attention file assumes finite arithmetic (-ffinite-math-only: no NaN/Inf handling is wanted anywhere in
# it, masks are finite sentinels): fmaxf on MFMA outputs then compiles to bare v_max3 without quieting moves.
# (Self-attention likewise: 1387 -> 1325 us at B=32, N=4096, d=40.)
PER_SOURCE_FLAGS = {"sta_selfattn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-ffinite-math-only"],
                    "sta_selfattn_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-ffinite-math-only"],
                    "sta_xattn.hip": ["-ffinite-math-only"], "sta_xattn_bwd.hip": ["-ffinite-math-only"], "sta_xattn_proj.hip": ["-ffinite-math-only"],
                    "sta_xattn_proj3.hip": ["-ffinite-math-only"], "sta_rowgemm.hip": ["-ffinite-math-only"], "sta_ffgemm.hip": ["-ffinite-math-only"], "sta_conv.hip": ["-ffinite-math-only"], "sta_gemm.hip": ["-ffinite-math-only"]}

STA_BF16, STA_F16 = 0, 1
OPT_FWD_KERNEL, OPT_STAGED_TILES, OPT_STAGED_WAVES, OPT_STAGED_QT, OPT_HEAD_MAJOR, OPT_SPLIT_QT, OPT_SELFATTN_32, OPT_PROJ_PAIR, OPT_SELFATTN_WAVES, OPT_SELFATTN_PIPE, OPT_PROJ_LL2, OPT_BWD_KERNEL, OPT_BWD_SLOTS, OPT_BWD_WAVES = range(14)
FWD_STAGED, FWD_SPLIT = 1, 2
MAX_KEYS, MAX_HEAD_DIM, MAX_OBJECTS = 80, 160, 8
P3_STATS_WORDS = 8

# every symbol include/sta_xattn.h and include/sta_unet.h declare: (restype, argtypes)
_vp, _i, _f, _sz, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_long
SYMBOLS = {
    "sta_version": (_i, []),
    "sta_last_error": (ctypes.c_char_p, []),
    "sta_built_with": (ctypes.c_char_p, []),
    "sta_set_option": (_i, [_i, _i]),
    "sta_xattn_packed_kv_bytes": (_sz, [_i, _i, _i]),
    "sta_xattn_pack_kv": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sta_xattn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "sta_xattn_fwd_proj_supported": (_i, [_i, _i, _i, _i]),
    "sta_xattn_fwd_proj_locals_from_l2": (_i, [_i, _i, _i, _i]),
    "sta_xattn_packed_wq_bytes": (_sz, [_i, _i]),
    "sta_xattn_pack_wq": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sta_xattn_packed_kv_proj_bytes": (_sz, [_i, _i, _i]),
    "sta_xattn_pack_kv_proj": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sta_xattn_fwd_proj": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "sta_xattn_fwd_proj_qfrag_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "sta_xattn_fwd_proj_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp, _vp]),
    "sta_xattn_fwd_proj_qfrag": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "sta_xattn_fwd_proj_qfrag_ofrag": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "sta_to_out_ln_packed_wo_bytes": (_sz, [_i, _i]),
    "sta_to_out_ln_pack_wo": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sta_to_out_ln_ofrag": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _f, _i, _i, _vp]),
    "sta_selfattn_fwd_sfrag": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _l, _l, _f, _i, _vp]),
    "sta_xattn_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "sta_xattn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "sta_selfattn_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _l, _l, _f, _i, _vp]),
    "sta_selfattn_fwd_optimistic": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _l, _l, _f, _i, _i, _vp]),
    "sta_selfattn_optimistic_supported": (_i, [_i, _i, _i, _f, _i]),
    "sta_selfattn_optimistic_flags_bytes": (ctypes.c_size_t, [_i, _i, _i]),
    "sta_selfattn_fwd_lse": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _l, _l, _f, _i, _vp]),
    "sta_selfattn_bwd": (_i, [_vp] * 13 + [_i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "sta_groupnorm_silu": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "sta_geglu": (_i, [_vp, _vp, _l, _i, _i, _vp]),
    "sta_add_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _i, _vp]),
    "sta_add_layernorm_qfrag": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _i, _vp]),
    "sta_ff_geglu_packed_w_bytes": (_sz, [_i, _i]),
    "sta_ff_geglu_pack_w": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sta_ff_geglu_qfrag": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _vp]),
    "sta_ff_out_packed_w_bytes": (_sz, [_i, _i]),
    "sta_ff_out_pack_w": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sta_ff_out_res_hfrag": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp]),
    "sta_conv3x3_nhwc_supported": (_i, [_i, _i, _i, _i, _i]),
    "sta_conv3x3_packed_w_bytes": (_sz, [_i, _i]),
    "sta_conv3x3_pack_w": (_i, [_vp, _l, _l, _l, _l, _vp, _i, _i, _i, _vp]),
    "sta_conv3x3_nhwc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sta_linear_rows_supported": (_i, [_l, _i, _i]),
    "sta_linear_rows_packed_w_bytes": (_sz, [_i, _i]),
    "sta_linear_rows_pack_w": (_i, [_vp, _l, _l, _vp, _i, _i, _i, _vp]),
    "sta_linear_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp]),
    "sta_conv3x3_stats_slots": (_i, [_i, _i]),
    "sta_stats_finalize": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sta_linear_rows_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _i, _i, _i, _vp]),
    "sta_groupnorm_silu_nhwc_cstats": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "sta_linear_rows_cat": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp]),
    "sta_groupnorm_silu_nhwc_cat": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "sta_add_bias_nchw": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sta_groupnorm_nhwc_workspace_bytes": (_sz, [_i, _i, _i]),
    "sta_groupnorm_silu_nhwc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "sta_add_bias_rows": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _vp]),
    "sta_quant_rows_fp8": (_i, [_vp, _vp, _vp, _l, _i, _i, _vp]),
    "sta_groupnorm_silu_nhwc_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "sta_geglu_bwd": (_i, [_vp, _vp, _vp, _l, _i, _i, _vp]),
    "sta_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _f, _i, _vp]),
}


class StaLibraryError(RuntimeError):
    pass


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = SOURCES + [os.path.join(_HERE, "isa_lint.py"), os.path.join(INCLUDE, "sta_xattn.h"), os.path.join(INCLUDE, "sta_unet.h"), os.path.join(CSRC, "sta_internal.h"), os.path.join(CSRC, "sta_xattn_dev.h"), os.path.join(CSRC, "sta_xattn_proj3.h"), os.path.join(CSRC, "sta_selfattn_dev.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


LINT_LOG = os.path.join(CSRC, ".isa_lint.log")


def _llvm_bin(hipcc):
    """Directory of the clang / lld / clang-offload-bundler that belong to `hipcc` (the same ROCm prefix: no mixed toolchain)."""
    for root in (os.environ.get("ROCM_PATH"), os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "/opt/rocm"):
        if root and os.path.exists(os.path.join(root, "lib", "llvm", "bin", "clang")):
            return os.path.join(root, "lib", "llvm", "bin")
    raise StaLibraryError("no lib/llvm/bin/clang beside %s (set ROCM_PATH)" % hipcc)


def _hipcc_headline(hipcc):
    ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip().splitlines()
    return ver[0] if ver else "hipcc ?"


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise StaLibraryError("build step failed: %s\n%s" % (" ".join(cmd), r.stdout + r.stderr))


def _compile_checked(hipcc, base, flags, src, obj, workdir, verbose):
    """One translation unit: device code to ASSEMBLY, the hazard lint + its s_nop cure on that text (sta/isa_lint.py: the table
    measured by tools/hazard_probe.py, which knows one pair hipcc 7.2 does not pad — an MFMA accumulating into the result of an MFMA
    of another shape), then assembler -> lld -> bundle -> host compile with that device image: the steps `hipcc -c` runs itself,
    with the text pass in the middle. -> [(line, wait states inserted, rule)]"""
    from . import isa_lint
    _LLVM_BIN = _llvm_bin(hipcc)
    stem = os.path.join(workdir, os.path.basename(src))
    if verbose:
        print(" ".join(base + flags + ["--cuda-device-only", "-S", src]))
    _run(base + flags + ["--cuda-device-only", "-S", src, "-o", stem + ".s"])
    with open(stem + ".s") as fh:
        text = fh.read()
    fixed, log = isa_lint.fix_text(text)
    left = isa_lint.lint_text(fixed)
    if left:
        raise StaLibraryError("hazard lint of %s not clean after the s_nop pass:\n%s" % (src, isa_lint.format_findings(left)))
    with open(stem + ".fixed.s", "w") as fh:
        fh.write(fixed)
    _run([_LLVM_BIN + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", stem + ".fixed.s", "-o", stem + ".dev.o"])
    _run([_LLVM_BIN + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", stem + ".dev.o", "-o", stem + ".dev.out"])
    _run([_LLVM_BIN + "/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
          "-input=/dev/null", "-input=" + stem + ".dev.out", "-output=" + stem + ".hipfb"])
    host_defs = ['-DSTA_BUILT_WITH="%s"' % _hipcc_headline(hipcc).replace('"', "'")] if os.path.basename(src) == "sta_xattn.hip" else []
    _run(base + flags + host_defs + ["--cuda-host-only", "-c", src, "-Xclang", "-fcuda-include-gpubinary", "-Xclang", stem + ".hipfb", "-o", obj])
    return log


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into csrc/libsta_xattn.so (cross-compiles without a GPU). Every kernel's assembly passes the
    hazard lint (and its deterministic cure) on the way: csrc/.isa_lint.log lists what was padded."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise StaLibraryError("hipcc not found; cannot build %s" % LIB_PATH)
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
            *os.environ.get("STA_HIPCC_FLAGS", "").split()]           # env: experiments only (tools/)
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    workdir = tempfile.mkdtemp(prefix="sta_build_")
    objs = [os.path.join(workdir, os.path.basename(src) + ".o") for src in SOURCES]
    try:
        with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:      # one pipeline per source, side by side
            futs = [ex.submit(_compile_checked, hipcc, base, PER_SOURCE_FLAGS.get(os.path.basename(src), []), src, obj, workdir, verbose)
                    for src, obj in zip(SOURCES, objs)]
            logs = [f.result() for f in futs]
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH + ".tmp"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise StaLibraryError("link failed:\n" + r.stdout + r.stderr)
        os.replace(LIB_PATH + ".tmp", LIB_PATH)
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
    with open(LINT_LOG, "w") as fh:
        fh.write("# %s\n" % _hipcc_headline(hipcc))
        for src, log in zip(SOURCES, logs):
            fh.write("%s: %d site(s) padded\n" % (os.path.basename(src), len(log)))
            for ln, states, rule in sorted(log):
                fh.write("    line %d: +%d wait state(s): %s\n" % (ln, states, rule))
    if verbose:
        print(open(LINT_LOG).read())
    return LIB_PATH


# The hipcc release whose output the GPU parity suite ran against (the hazard windows sta/isa_lint.py enforces were measured with it,
# tools/hazard_probe.py). A library built by another release still passes the lint, but its hot kernel is checked numerically on first use
# (sta.ops.toolchain_self_check) instead of being trusted.
VALIDATED_HIPCC = "HIP version: 7.2."


def built_with():
    """The `hipcc --version` headline of the build that produced the LOADED library, read back from the library itself (sta_built_with;
    csrc/.isa_lint.log — what the lint padded — is a by-product of the build, not tracked and not what is trusted). '' if unknown."""
    try:
        return load().sta_built_with().decode()
    except (StaLibraryError, AttributeError):
        return ""


def toolchain_validated(headline=None):
    return (built_with() if headline is None else headline).startswith(VALIDATED_HIPCC)


_lib = None


def load():
    """Load the library and bind every declared symbol. Raises StaLibraryError if impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StaLibraryError(
            "%s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU or eager fallback for the fused cross-attention)" % LIB_PATH)
    # PyTorch ships its own HIP runtime; the library must bind to the one torch initialised (the tensors it is handed live
    # there). Loaded the other way round — libsta first, then torch — the process ends up with two runtimes and every launch
    # of this library fails with "no ROCm-capable device is detected". Importing torch first pins the order.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise StaLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise StaLibraryError("%s does not export %s" % (LIB_PATH, name)) from e
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


_options = {}


def set_option(key, value):
    """sta_set_option: override a launch heuristic (tests / tools only; 0 = automatic)."""
    check(load().sta_set_option(int(key), int(value)), "sta_set_option")
    _options[int(key)] = int(value)


def get_option(key):
    """The value this process last gave sta_set_option for `key` (0 = automatic / never set; the C-ABI has no getter)."""
    return _options.get(int(key), 0)


def last_error():
    return load().sta_last_error().decode()


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, last_error()))
