"""Static checks on the gfx950 code hipcc generates for the hot kernels (no GPU needed; hipcc cross-compiles).  A spill or a
register count past the occupancy the kernel was designed for does not break parity - it silently costs tens of percent - so
it is pinned here, together with the "load inside a bounds guard" trap (tools/isa_lint.py)."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chronoedit_amd", "csrc")
spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
isa_lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_lint)

HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _rows(name):
    # (kernel, insts, regs, scratch, mfma, serialised loads, s_nop, v_mov)
    return isa_lint.lint(os.path.join(CSRC, name), 60)


def _pick(rows, *needles):
    hit = [r for r in rows if all(n in r[0] for n in needles)]
    assert hit, (needles, [r[0] for r in rows])
    return hit


def test_attention_kernels_fit_two_waves_per_simd_without_spills():
    rows = _rows("ce_attn.hip")
    for r in _pick(rows, "attn_fwd_sp_kernel"):  # the default loop body, one and two KV segments
        assert r[3] == 0, f"{r[0]}: {r[3]} bytes of scratch"
        assert r[2] <= 256, f"{r[0]}: {r[2]} registers (512 threads per workgroup = two waves per SIMD)"
        assert r[5] == 0, f"{r[0]}: a load is waited for right where it is issued"
    for r in rows:
        assert r[3] == 0, f"{r[0]}: scratch"


def test_gemm_kernels_have_no_spills_and_a_batched_epilogue():
    for f in ("ce_gemm256.hip", "ce_gemm_fp8.hip"):
        rows = _rows(f)
        for r in rows:
            assert r[3] == 0, f"{r[0]}: scratch"
            if "gemm_bf16_256" in r[0] or "gemm_fp8_256" in r[0]:
                assert r[2] <= 256, (r[0], r[2])
                # bias / scale loads of the first epilogue half only; the residual / gate loads are batched (ce_gemm_epi.h)
                assert r[5] <= 8, f"{r[0]}: {r[5]} serialised loads"


def test_one_wave_per_simd_gemm_keeps_its_accumulators_in_place():
    """ce_gemm256w4.hip: 256 accumulator AGPRs + the fragment sets in VGPRs, no scratch; per trip of the K loop (two K-tiles) exactly
    256 MFMAs, 64 fragment reads, 32 LDS-DMA pieces, 4 barriers (2 in the one-barrier form) and NOT ONE v_accvgpr_* - as builtins the
    MFMAs came out untied and hipcc shuffled every result through VGPRs (hundreds of v_accvgpr_* per K-tile): the trap this pins."""
    src = os.path.join(CSRC, "ce_gemm256w4.hip")
    for r in _rows("ce_gemm256w4.hip"):
        assert r[3] == 0, f"{r[0]}: scratch"
        assert r[2] <= 512, r
    loops = isa_lint.inner_loops(src, "gemm_bf16_w4")
    # 5 epilogues x (3 stages, 3 stages + one barrier, 2 stages) + the two instantiations with a two-level segmented A operand (the
    # convolutions of the VAE: ce_conv3d_gemm_bf16), each also as the 256 x 128 tile (NG = 4) and the 256 x 96 tile (NG = 3, round 5) -
    # template arguments <EPI, NSA, ONEBAR, SEG2, NG>
    assert len(loops) == 21
    for name, c in loops:
        ng = 4 if "ELi4EEE" in name else 3 if "ELi3EEE" in name else 8  # column fragments per wave: MFMAs and W pieces scale with it, 8 + NG fragments per k-step
        assert c.get("v_mfma_f32_16x16x32_bf16", 0) == 32 * ng, (name, c)
        assert c.get("ds_read_b128", 0) == 4 * (8 + ng) and c.get("buffer_load_dwordx4", 0) == 2 * (8 + ng), (name, c)
        assert c.get("s_barrier", 0) == (2 if re.search(r"ELi\dELb1ELb", name) else 4), (name, c)
        assert not any(op.startswith("v_accvgpr") or op.startswith("scratch") for op in c), (name, c)


def test_tile384_gemm_keeps_its_384_accumulators_in_place():
    """ce_gemm384.hip: 384 accumulator registers per lane (256 AGPRs + 128 VGPRs) out of 512; per trip of the K loop (two K-tiles of
    a 192x128 wave tile) exactly 384 MFMAs, 80 fragment reads (48 A through the ring + 32 W), 40 LDS-DMA pieces and ONE barrier per
    K-tile; nothing moves between register files and nothing spills inside the loop (the gate + residual epilogue parks a few
    registers in scratch AFTER the loop: bounded here)."""
    src = os.path.join(CSRC, "ce_gemm384.hip")
    # two instantiations per epilogue (round 6): NF = 12 row fragments per wave (384 x 256) and NF = 9 (288 x 256: the same loop, 18 groups per K-tile)
    for r in _pick(_rows("ce_gemm384.hip"), "gemm_bf16_384"):
        nf = 9 if "ELi9EEE" in r[0] else 12
        assert r[2] <= 512 and r[4] == 32 * nf, r
        assert r[3] <= (128 if "ILi2E" in r[0] else 0), f"{r[0]}: scratch"
    loops = isa_lint.inner_loops(src, "gemm_bf16_384")
    assert len(loops) == 12  # five epilogues + the transposed store (EPI_BIAS_T), two tile heights each
    for name, c in loops:
        nf = 9 if "ELi9EEE" in name else 12
        assert c.get("v_mfma_f32_16x16x32_bf16", 0) == 32 * nf, (name, c)
        assert c.get("ds_read_b128", 0) == 2 * (2 * nf + 16) and c.get("buffer_load_dwordx4", 0) == 2 * (nf + 8), (name, c)
        assert c.get("s_barrier", 0) == 2, (name, c)
        assert not any(op.startswith("v_accvgpr") or op.startswith("scratch") for op in c), (name, c)


def test_fp8_one_wave_per_simd_gemm_loop():
    """ce_gemm_fp8w4.hip: per trip of the K loop (two K-tiles of 128 bytes) 128 MX MFMAs on 8-register fragments, 64 fragment reads (two
    16-byte halves each), 32 LDS-DMA pieces, ONE barrier per K-tile; accumulators tied in the AGPRs, nothing spilled in the loop."""
    src = os.path.join(CSRC, "ce_gemm_fp8w4.hip")
    for r in _pick(_rows("ce_gemm_fp8w4.hip"), "gemm_fp8_w4"):
        assert r[2] <= 512 and r[3] <= 16, r
    loops = isa_lint.inner_loops(src, "gemm_fp8_w4")
    # template arguments <EPI, MX, GP>: three epilogues x {per-row scales: the plain instruction | MX: the block-scaled one, with the gated
    # residual's prefetching epilogue (GP) or the generic per-pass one} + the fused bias + GELU + MX-quantising epilogue (EPI 7) of FFN-up (32- and 64-bit store offsets)
    assert len(loops) == 11
    for name, c in loops:
        mfma = c.get("v_mfma_f32_16x16x128_f8f6f4", 0) + c.get("v_mfma_scale_f32_16x16x128_f8f6f4", 0)
        assert mfma == 128 and (c.get("v_mfma_f32_16x16x128_f8f6f4", 0) == 0 or c.get("v_mfma_scale_f32_16x16x128_f8f6f4", 0) == 0), (name, c)
        assert c.get("ds_read_b128", 0) == 64 and c.get("buffer_load_dwordx4", 0) == 32 and c.get("s_barrier", 0) == 2, (name, c)
        assert not any(op.startswith("v_accvgpr") or op.startswith("scratch") for op in c), (name, c)
        if c.get("v_mfma_scale_f32_16x16x128_f8f6f4", 0):  # MX: the block scales of a K-tile = two 8-byte loads per lane, nothing else
            assert c.get("global_load_dwordx2", 0) == 4 and c.get("v_mov_b32_e32", 0) <= 2, (name, c)
            assert not any("vmcnt" in op for op in c if op.startswith("s_waitcnt") and op not in ("s_waitcnt",)), (name, c)


def test_row_kernels_issue_their_row_loads_back_to_back():
    rows = _rows("ce_rowops.hip")
    (rr,) = _pick(rows, "rmsnorm_rope_kernel", "ILb1E")  # FULL variant (D = 5120)
    assert rr[5] == 0 and rr[3] == 0 and rr[2] <= 168, rr  # three waves per SIMD
    for r in _pick(rows, "ln_affine_kernel", "ELb1ELi1E"):  # FULL variants, one row per wave (fp8 per-row and MX outputs; bf16 fallback)
        assert r[3] == 0 and r[2] <= 168, r
        assert r[5] <= 20, r  # the (a, b) table reads of the third pass (L2 hits); the ten row loads are not among them
    (two,) = _pick(rows, "ln_affine_kernel", "ILi0ELb1ELi2E")  # bf16 (FP8 = 0), two rows per wave sharing the (a, b) chunks
    assert two[3] == 0 and two[2] <= 256 and two[5] == 0, two


def test_mxfp8_attention_kernels_have_no_spills_and_their_matrix_work_per_tile():
    rows = _rows("ce_attn_fp8.hip")
    for r in rows:
        if "attn_fwd_mxfp8_sp_kernel" not in r[0]:
            assert r[3] == 0, f"{r[0]}: scratch"
    (sp,) = _pick(rows, "attn_fwd_mxfp8_sp_kernel")
    # the persistent item loop parks a few lane-invariant registers in scratch at kernel entry and reloads them once per work item
    # (prologue / epilogue blocks only; the key-tile loop has no scratch access - checked on the assembly below)
    assert sp[3] <= 128, sp
    asm_loop_clean = isa_lint.loop_scratch_free(os.path.join(CSRC, "ce_attn_fp8.hip"), "attn_fwd_mxfp8_sp_kernel")
    assert asm_loop_clean, "scratch access inside a loop block that issues MFMAs"
    assert sp[2] <= 256, sp  # 512 threads per workgroup = two waves per SIMD
    assert sp[5] <= 1, sp    # no tile load is waited for where it is issued (LDS-DMA three iterations ahead); one per-item prologue load is
    # 4 S + 4 P.V + 1 row-sum MFMA in the steady-state body, the peeled first tile (4 + 1), the repair route (4) and the drain (4 + 1 ...)
    assert 20 <= sp[4] <= 28, sp


def test_vae_head_conv_and_mid_block_attention_kernels_keep_their_shape():
    """ce_conv.hip, round 4.  conv_head_kernel: 90 MFMAs per kt (10 input rows x 9 fragments) in three workgroups' worth of registers, its
    input fragments by BUFFER loads issued a row ahead (a frame pointer read back from LDS made hipcc emit flat loads with a full wait
    behind every one: 10 x slower) - no flat load, no load waited for where it is issued beyond the weight staging.  attn_1head_kernel
    (round 6 form: LDS-DMA ring, two query blocks per wave): no scratch, per 32-key tile 96 MFMAs on 48 fragment reads and 12 DMA pieces behind
    one barrier; the accumulators live in the AGPRs as the MFMAs' own operand: per tile only the conditional rescale blocks move them."""
    import subprocess
    import tempfile
    rows = _rows("ce_conv.hip")
    (head,) = _pick(rows, "conv_head_kernel")
    assert head[3] == 0 and head[2] <= 168 and head[4] == 90 and head[5] <= 1, head
    for r in _pick(rows, "attn_1head_kernel"):
        assert r[3] == 0, r
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-S", "--cuda-device-only", os.path.join(CSRC, "ce_conv.hip"),
                        "-o", asm], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()

    def body(sub):
        m = re.search(r"^(\S*" + sub + r"\S*):\s*; @\1$", text, flags=re.M)
        return text[m.end():text.index(".Lfunc_end", m.end())]

    hb = body("conv_head_kernel")
    assert "flat_load" not in hb and hb.count("buffer_load_dwordx4") >= 90, (hb.count("flat_load"), hb.count("buffer_load_dwordx4"))
    ab = body("attn_1head_kernelILi384E")
    assert "flat_load" not in ab and "ds_write" not in ab and "scratch_" not in ab
    # round 6: K / V^T tiles by LDS-DMA (no staging registers): 2 prologue tiles + the in-loop run-ahead tile, 12 pieces each; Q by 24 plain loads
    assert ab.count("buffer_load_dwordx4") == 3 * 12 and ab.count(" lds") == 3 * 12 and ab.count("global_load_dwordx4") == 24, ab.count(" lds")
    (loop,) = isa_lint.inner_loops(os.path.join(CSRC, "ce_conv.hip"), "attn_1head_kernelILi384E")
    c = loop[1]  # per 32-key tile: 2 query blocks x (24 score + 24 P.V) products on 48 fragment reads, 12 DMA pieces, ONE barrier
    assert c.get("v_mfma_f32_16x16x32_bf16", 0) == 96 and c.get("ds_read_b128", 0) == 48 and c.get("buffer_load_dwordx4", 0) == 12, c
    assert c.get("s_barrier", 0) == 1 and not any(op.startswith("scratch") or op.startswith("ds_write") for op in c), c
    # the accumulators and the first block's Q live in the accumulator file as the MFMAs' own operands: only the two conditional rescale
    # blocks (2 x 96 registers out and back) move them inside the loop
    assert c.get("v_accvgpr_read_b32", 0) <= 192 + 8 and c.get("v_accvgpr_write_b32", 0) <= 192 + 8, c


def test_vae_slab_conv_kernels_keep_their_shape():
    """ce_conv.hip, round 6: conv3x3_c96 (the 96-channel full-resolution convs with the input slab in the LDS).  Per sub-stage (one (kt, kh)
    row of 32 input channels) a wave of the two-waves-per-SIMD form issues 72 MFMAs on 12 A + 18 W fragment reads and 7 LDS-DMA pieces behind ONE
    barrier (one wave per SIMD: 144 / 24 + 18 / 13); accumulators tied in the AGPRs, no scratch, no ds_write, no accumulator moves in the loop."""
    rows = _rows("ce_conv.hip")
    for r in _pick(rows, "conv3x3_c96_w8_kernel"):
        assert r[3] == 0 and r[2] <= 256, r   # two waves per SIMD
    for r in _pick(rows, "conv3x3_c96_kernel"):
        assert r[3] == 0 and r[2] <= 512, r
    for sub, mfma, reads, pieces in (("conv3x3_c96_w8_kernelILi3E", 72, 30, 7), ("conv3x3_c96_w8_kernelILi6E", 72, 30, 7), ("conv3x3_c96_kernelILi3E", 144, 42, 13)):
        (loop,) = isa_lint.inner_loops(os.path.join(CSRC, "ce_conv.hip"), sub)
        c = loop[1]
        assert c.get("v_mfma_f32_16x16x32_bf16", 0) == mfma and c.get("ds_read_b128", 0) == reads and c.get("buffer_load_dwordx4", 0) == pieces, (sub, c)
        assert c.get("s_barrier", 0) == 1, (sub, c)
        assert not any(op.startswith("scratch") or op.startswith("ds_write") or op.startswith("v_accvgpr") for op in c), (sub, c)

