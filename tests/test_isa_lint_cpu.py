"""The offline hazard lint (sta/isa_lint.py) that sta.lib.build() runs over every kernel's assembly: its rules on hand-written
snippets (each pair at the distance tools/hazard_probe.py measured as failing on MI355X, and one state beyond it), its s_nop cure,
and the hazard-sensitive translation unit of the library compiled here (hipcc cross-compiles without a GPU)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import isa_lint, lib  # noqa: E402


def kernel(*body):
    return "\t.text\nk_test:\n" + "\n".join("\t" + b for b in body) + "\n\ts_endpgm\n.Lfunc_end0:\n"


M32 = "v_mfma_f32_16x16x32_bf16 v[50:53], v[10:13], v[18:21], v[30:33]"
M16_ACC = "v_mfma_f32_16x16x16_bf16 v[50:53], v[10:11], v[18:19], v[50:53]"
NOP = "s_nop 0"


def rules(text):
    return [f.rule for f in isa_lint.lint_text(text)]


def test_mixed_shape_accumulate_chain_is_flagged_until_five_states_or_a_third_mfma():
    # measured: wrong for 0..4 wait states of s_nop / scalar fillers (registers 0 and 1 of the tile), right from 5; the lint asks 6
    for k in range(6):
        assert "mfma D -> C of an mfma of another shape" in rules(kernel(M32, *[NOP] * k, M16_ACC)), k
    assert rules(kernel(M32, *[NOP] * 6, M16_ACC)) == []
    assert rules(kernel(M32, "s_nop 5", M16_ACC)) == []
    # one independent MFMA between the two serialises them on the matrix pipe: measured right at distance 1
    assert rules(kernel(M32, "v_mfma_f32_16x16x32_bf16 v[60:63], v[10:13], v[18:21], v[60:63]", M16_ACC)) == []
    # the same shape back to back is the ordinary accumulate chain: interlocked
    assert rules(kernel(M32, "v_mfma_f32_16x16x32_bf16 v[50:53], v[14:17], v[22:25], v[50:53]")) == []
    # the reverse order and another destination are the same hazard
    assert rules(kernel("v_mfma_f32_16x16x16_bf16 v[50:53], v[10:11], v[18:19], v[30:33]", "v_xor_b32_e32 v1, 1, v2",
                        "v_mfma_f32_16x16x32_bf16 v[60:63], v[10:13], v[18:21], v[50:53]")) == ["mfma D -> C of an mfma of another shape"]


def test_result_read_and_write_windows():
    rd = "v_add_f32_e32 v1, v50, v51"
    assert "mfma D -> read" in rules(kernel(M32, *[NOP] * 7, rd))
    assert rules(kernel(M32, *[NOP] * 8, rd)) == []
    assert "mfma D -> read" in rules(kernel(M32, "s_nop 3", "global_store_dwordx4 v1, v[50:53], s[2:3]"))
    wr = "v_mov_b32_e32 v52, v1"
    assert "mfma D -> write" in rules(kernel(M32, *[NOP] * 4, wr))
    assert rules(kernel(M32, *[NOP] * 5, wr)) == []
    big = "v_mfma_f32_32x32x16_bf16 v[50:65], v[10:13], v[18:21], v[30:45]"
    assert "mfma D -> read" in rules(kernel(big, "s_nop 10", rd))
    assert rules(kernel(big, "s_nop 11", rd)) == []
    # C of the 8-pass shape is read one pass per four registers: registers 4.. may not be written for 4 states; 16x16 shapes: never flagged
    assert "mfma reads C -> write" in rules(kernel(big, *[NOP] * 3, "v_mov_b32_e32 v45, v1"))
    assert rules(kernel(big, *[NOP] * 3, "v_mov_b32_e32 v31, v1")) == []
    assert rules(kernel(M32, "v_mov_b32_e32 v33, v1")) == []
    assert rules(kernel(M32, "v_mov_b32_e32 v13, v1")) == []          # A / B: read at issue (probed also behind a queue of MFMAs)
    # an MFMA that takes the result as A or B waits like a vector read; as C of the same shape it does not
    assert "mfma D -> read" in rules(kernel(M32, "s_nop 3", "v_mfma_f32_16x16x32_bf16 v[60:63], v[50:53], v[18:21], v[60:63]"))


def test_vector_producers():
    mv = "v_mov_b32_e32 v10, v1"
    assert "vector write -> mfma operand" in rules(kernel(mv, NOP, M32))
    assert rules(kernel(mv, NOP, NOP, M32)) == []
    assert "vector write -> permlane swap" in rules(kernel(mv, NOP, "v_permlane32_swap_b32 v10, v11"))
    assert rules(kernel(mv, "s_nop 1", "v_permlane16_swap_b32 v11, v10")) == []
    assert "vector write -> dpp source" in rules(kernel(mv, "v_mov_b32_dpp v3, v10 row_ror:8 row_mask:0xf bank_mask:0xc"))
    assert "transcendental -> vector read" in rules(kernel("v_rcp_f32_e32 v5, v6", "v_mul_f32_e32 v7, v5, v5"))
    assert rules(kernel("v_rcp_f32_e32 v5, v6", NOP, "v_mul_f32_e32 v7, v5, v5")) == []


def test_instructions_inside_asm_statements_count_and_are_checked():
    # hipcc counts an asm statement as zero wait states and pads nothing inside it; the lint sees plain instructions
    body = [M32, ";;#ASMSTART", "s_nop 1", "v_mov_b32 v50, v1", ";;#ASMEND"]
    assert "mfma D -> write" in rules(kernel(*body))
    body = [M32, ";;#ASMSTART", "s_nop 7", "v_mov_b32 v1, v50", ";;#ASMEND"]
    assert rules(kernel(*body)) == []


def test_paths_follow_branches_and_loop_back_edges():
    text = kernel("s_cbranch_scc1 .LBB0_2", M32, "s_branch .LBB0_3", ".LBB0_2:", "s_nop 7", ".LBB0_3:", "v_add_f32_e32 v1, v50, v51")
    assert rules(text) == ["mfma D -> read"]                       # only the path through the MFMA is short
    loop = kernel(".LBB0_1:", "v_add_f32_e32 v1, v50, v51", "s_nop 7", M32, "s_cbranch_scc1 .LBB0_1")
    assert rules(loop) == ["mfma D -> read"]                       # consumer reached over the back edge


def test_fix_pads_exactly_the_deficit_and_converges():
    text = kernel(M32, "v_xor_b32_e32 v1, 1, v2", M16_ACC, "s_nop 1", "v_add_f32_e32 v3, v50, v51")
    fixed, log = isa_lint.fix_text(text)
    assert isa_lint.lint_text(fixed) == []
    assert sorted((n, r) for _, n, r in log) == [(5, "mfma D -> C of an mfma of another shape"), (5, "mfma D -> read")]
    assert fixed.count("s_nop 4") == 2 and "sta isa_lint" in fixed
    clean = kernel(M32, "s_nop 7", "v_add_f32_e32 v3, v50, v51")
    assert isa_lint.fix_text(clean) == (clean, [])


def test_hazard_sensitive_translation_unit_is_clean_after_the_pass():
    """csrc/sta_xattn_proj3.hip as build() treats it: whatever this hipcc leaves unpadded is padded, nothing else remains."""
    src = os.path.join(lib.CSRC, "sta_xattn_proj3.hip")
    path = isa_lint.compile_to_asm(src, lib.PER_SOURCE_FLAGS["sta_xattn_proj3.hip"], [lib.INCLUDE, lib.CSRC])
    text = open(path).read()
    assert text.count("v_mfma_f32_16x16x32_bf16") > 100            # every instantiation is in the text, the bf16 out-fragment one too
    assert "xattn_fwd_proj_p3_kernelIDF16bLi10ELi2ELb1" in text
    fixed, log = isa_lint.fix_text(text)
    assert isa_lint.lint_text(fixed) == []
    assert {r for _, _, r in log} <= {"mfma D -> C of an mfma of another shape"}      # hipcc's own table covers every other pair


def test_built_library_carries_its_lint_log():
    lib.build()
    log = open(lib.LINT_LOG).read()
    for src in lib.SOURCES:
        assert os.path.basename(src) + ":" in log


def test_build_records_its_toolchain_and_the_validated_release_is_recognised():
    """The library carries the `hipcc --version` headline of its own build (sta_built_with, read back by sta.lib.built_with — not a file
    beside it); a library built by the release the GPU parity suite ran against counts as validated, any other one makes sta.ops run
    toolchain_self_check before a block decides on fragment layouts (advisor items of rounds 4 and 5)."""
    from sta import lib
    lib.build()
    assert lib.built_with().startswith("HIP version: ") and lib.toolchain_validated()
    assert open(lib.LINT_LOG).readline().lstrip("# ").strip() == lib.built_with()
    assert not lib.toolchain_validated("HIP version: 9.9.12345-deadbeef") and not lib.toolchain_validated("")
