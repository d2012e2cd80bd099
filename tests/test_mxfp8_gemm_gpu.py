"""The MX form of the fp8 GEMM path (round 4; BASELINE.json configs[4] "fp8 weights"): OCP MXFP8 operands - e4m3 elements, one E8M0 scale
per 32 consecutive K elements of a row - with the block scales applied inside the matrix pipe (v_mfma_scale_f32_16x16x128_f8f6f4).
  * ce_quant_rows_mxfp8 against the contract (oracle.dit_oracle.mx_quant): scale bytes and element bytes, exactly;
  * ce_ln_affine_mxfp8 == ce_ln_affine_bf16 followed by ce_quant_rows_mxfp8, bit for bit;
  * ce_gemm_mxfp8 against fp32 math on the dequantised operands - with block magnitudes spread over 2^-6 .. 2^6 inside every row, so that
    a scale byte applied to the wrong 32 elements (the lane geometry of profiles/r04_mx16_probe.txt) or in the wrong K-tile shows at once;
  * the quantisation error of MX against per-row scales on the same operands (reported; the reason for the form)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _spread(rows, K, g, lo=-6, hi=6):
    """bf16 matrix whose 32-element blocks differ in magnitude by powers of two (each block of each row its own 2^e)."""
    x = torch.randn(rows, K, generator=g)
    e = torch.randint(lo, hi + 1, (rows, K // 32), generator=g).float()
    return (x * torch.exp2(e).repeat_interleave(32, dim=1)).to(BF)


def _contract(x):
    """(E8M0 bytes [rows, K/32], e4m3 bytes [rows, K]) of the MX contract for a bf16 matrix."""
    xf = x.float()
    xb = xf.view(xf.shape[0], -1, 32)
    amax = xb.abs().amax(-1)
    e = torch.floor(torch.log2(torch.clamp(amax, min=2.0 ** -118))) - 8.0
    e = e + (amax > 448.0 * torch.exp2(e)).float()  # non-saturating: the smallest power of two with amax / scale <= 448
    e = torch.where(amax > 0, torch.clamp(e, min=-126.0), torch.full_like(e, -126.0))
    q = torch.clamp(xb / torch.exp2(e)[..., None], -448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).reshape(xf.shape)
    return (e + 127.0).to(torch.uint8), q


def _deq(q, sc_rows):
    return q.view(torch.float8_e4m3fn).float() * torch.exp2(sc_rows.float() - 127.0).repeat_interleave(32, dim=1)


@pytest.mark.parametrize("M,K", [(7, 256), (300, 5120), (1000, 13824), (129, 128)])
def test_quant_rows_mxfp8_matches_contract(M, K):
    from chronoedit_amd import ops
    g = torch.Generator().manual_seed(1)
    x = _spread(M, K, g)
    x[min(2, M - 1)] = 0          # an all-zero row: scale byte 1 (2^-126), elements 0
    x[0, 32:64] = 0               # an all-zero block inside a row
    q, s = ops.quant_rows_mxfp8(x.cuda())
    want_s, want_q = _contract(x)
    got_s = ops.mx_scales_to_rows(s, M, K).cpu()
    assert torch.equal(got_s, want_s), (got_s.int() - want_s.int()).abs().max()
    assert torch.equal(q.cpu(), want_q), (q.cpu().view(torch.float8_e4m3fn).float() - want_q.view(torch.float8_e4m3fn).float()).abs().max()
    assert q.cpu().view(torch.float8_e4m3fn).float().abs().max().item() <= 448.0
    assert rel_l2(_deq(q.cpu(), got_s), x) < 4e-2  # e4m3: 3 mantissa bits


def test_ln_affine_mxfp8_equals_two_launch_form():
    from chronoedit_amd import ops
    g = torch.Generator().manual_seed(4)
    for M, D in ((1003, 5120), (70, 1280)):
        x = (torch.randn(M, D, generator=g) * 2 + 0.3).to(BF).cuda()
        a = (1 + 0.2 * torch.randn(2, D, generator=g)).cuda()
        b = (0.1 * torch.randn(2, D, generator=g)).cuda()
        for kw in (dict(), dict(ab_rows=(M + 1) // 2, ab_stride=D)):
            aa, bb = (a, b) if kw else (a[0], b[0])
            h = ops.ln_affine(x, aa, bb, 1e-6, **kw)
            q_ref, s_ref = ops.quant_rows_mxfp8(h)
            q = torch.empty((M, D), dtype=torch.uint8, device="cuda")
            s = torch.zeros((ops.mx_scale_bytes(M, D),), dtype=torch.uint8, device="cuda")
            ops.ln_affine_mxfp8(x, aa, bb, 1e-6, out=q, scale=s, **kw)
            assert torch.equal(q, q_ref)
            assert torch.equal(ops.mx_scales_to_rows(s, M, D), ops.mx_scales_to_rows(s_ref, M, D))


@pytest.mark.parametrize("M,N,K,epi", [(256, 256, 256, "bias"), (300, 520, 512, "bias"), (1000, 1280, 5120, "gelu"), (7200, 5120, 13824, "gate"),
                                       (14400, 15360, 5120, "bias"), (13068, 5120, 5120, "gate"), (129, 264, 768, "bias")])
def test_gemm_mxfp8_matches_fp32_on_dequantised_operands(M, N, K, epi):
    from chronoedit_amd import ops
    g = torch.Generator().manual_seed(2)
    a = _spread(M, K, g, -4, 4).cuda()
    w = (_spread(N, K, g, -4, 4).float() * 0.03).to(BF).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(BF).cuda()
    gate = torch.randn(N, generator=g).cuda()
    aq, sa = ops.quant_rows_mxfp8(a)
    wq, sw = ops.quant_rows_mxfp8(w, w_order=True)
    kw = {"bias": dict(), "gelu": dict(epilogue=ops.EPI_BIAS_GELU), "gate": dict(epilogue=ops.EPI_GATE_RES, gate=gate, res=res)}[epi]
    out = ops.gemm_mxfp8(aq, sa, wq, sw, bias, **kw)
    ad = _deq(aq, ops.mx_scales_to_rows(sa, M, K))   # what the kernel is defined to compute, in fp32 on the GPU
    wd = _deq(wq, ops.mx_scales_to_rows(sw, N, K, w_order=True))
    lin = ad @ wd.t() + bias
    ref = {"bias": lin, "gelu": torch.nn.functional.gelu(lin.to(BF).float(), approximate="tanh"),
           "gate": res.float() + lin.to(BF).float() * gate}[epi]
    e = rel_l2(out, ref)
    assert e < 4e-3, e  # bf16 rounding of the output only
    # the price of fp8 under block scales vs under one scale per row, on these operands
    full = a.float() @ w.float().t() + bias
    e_mx = rel_l2(lin, full)
    qa, ra = ops.quant_rows_fp8(a)
    qw, rw = ops.quant_rows_fp8(w)
    lin_row = (qa.view(torch.float8_e4m3fn).float() * ra[:, None]) @ (qw.view(torch.float8_e4m3fn).float() * rw[:, None]).t() + bias
    e_row = rel_l2(lin_row, full)
    print(f"MX GEMM {M}x{N}x{K} {epi}: kernel-vs-definition {e:.2e}; quantisation error vs bf16 operands: MX blocks {e_mx:.2e}, per-row scales {e_row:.2e}")
    assert e_mx < 4e-2 and e_mx < 1.1 * e_row  # (e4m3's 3 mantissa bits set both; block scales must not be worse)


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (300, 512, 512), (7200, 13824, 5120), (1003, 1280, 5120)])
def test_gemm_mxfp8_gelu_quant_equals_gemm_then_quant(M, N, K):
    """The FFN-up form (ce_gemm_mxfp8_gelu_quant: bias + GELU epilogue that emits the next GEMM's MX operand) == ce_gemm_mxfp8 with the
    GELU epilogue followed by ce_quant_rows_mxfp8, bit for bit - element bytes and scale bytes - including the split-K tail of a partially
    filled last round (7200 x 13824: 6 tail tiles cut along K, summed by gemm_fp8w4_reduce_gelu_q)."""
    from chronoedit_amd import ops
    g = torch.Generator().manual_seed(6)
    a = _spread(M, K, g, -3, 3).cuda()
    w = (_spread(N, K, g, -3, 3).float() * 0.03).to(BF).cuda()
    bias = torch.randn(N, generator=g).cuda()
    aq, sa = ops.quant_rows_mxfp8(a)
    wq, sw = ops.quant_rows_mxfp8(w, w_order=True)
    h = ops.gemm_mxfp8(aq, sa, wq, sw, bias, epilogue=ops.EPI_BIAS_GELU)  # (both forms cut the same tail tiles along K and sum the slabs in the same order)
    q_ref, s_ref = ops.quant_rows_mxfp8(h)
    q = torch.empty((M, N), dtype=torch.uint8, device="cuda")
    s = torch.zeros((ops.mx_scale_bytes(M, N),), dtype=torch.uint8, device="cuda")
    ops.gemm_mxfp8_gelu_quant(aq, sa, wq, sw, bias, out=q, scale=s)
    assert torch.equal(q, q_ref), (q.view(torch.float8_e4m3fn).float() - q_ref.view(torch.float8_e4m3fn).float()).abs().max()
    assert torch.equal(ops.mx_scales_to_rows(s, M, N), ops.mx_scales_to_rows(s_ref, M, N))


@pytest.mark.parametrize("N,H,B", [(200, 2, 1), (333, 4, 2), (560, 8, 2), (1024, 2, 1)])
def test_mxfp8_attention_quantised_output_equals_attention_then_quant(N, H, B):
    """ce_attention_mxfp8_quant (the self-attention of the fp8 mode writing the out-projection's MX operand from its accumulators) ==
    ce_attention_mxfp8 followed by ce_quant_rows_mxfp8, bit for bit: element bytes and tiled scale bytes; remainder query blocks, several
    samples (the scale rows of sample b start at b Nq), the XCD-aware work order (H % 8 == 0)."""
    from chronoedit_amd import ops
    D = H * 128
    g = torch.Generator().manual_seed(N + H)
    qkv = torch.randn(B * N, 3 * D, generator=g).to(BF).cuda()
    one = torch.ones(D).cuda()
    q8, sq = ops.rmsnorm_rope_mxfp8(qkv[:, :D], one, None, 128, 1e-6, post_scale=ops.MXFP8_Q_SCALE)
    k8, sk = ops.rmsnorm_rope_mxfp8(qkv[:, D:2 * D], one, None, 128, 1e-6)
    v8t, sv = ops.v_mxfp8_transpose(qkv[:, 2 * D:], N, B, H)
    o = ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, batch=B)
    q_ref, s_ref = ops.quant_rows_mxfp8(o)
    q = torch.zeros((B * N, D), dtype=torch.uint8, device="cuda")
    s = torch.zeros((ops.mx_scale_bytes(B * N, D),), dtype=torch.uint8, device="cuda")
    ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, batch=B, out8=q, scale8=s)
    assert torch.equal(q, q_ref), (q.view(torch.float8_e4m3fn).float() - q_ref.view(torch.float8_e4m3fn).float()).abs().max()
    assert torch.equal(ops.mx_scales_to_rows(s, B * N, D), ops.mx_scales_to_rows(s_ref, B * N, D))


@pytest.mark.parametrize("Nq,H,B,Tt,Ti", [(200, 2, 1, 64, 33), (520, 4, 2, 512, 257), (1000, 8, 2, 128, 257)])
def test_cross_attention_quantised_output_equals_attention_then_quant(Nq, H, B, Tt, Ti):
    """ce_attention_2seg_vt_quant_bf16 (the cross-attention of the fp8 mode writing the out-projection's MX operand) ==
    ce_attention_2seg_vt_bf16 followed by ce_quant_rows_mxfp8, bit for bit."""
    from chronoedit_amd import ops
    D = H * 128
    g = torch.Generator().manual_seed(Nq + Ti)
    c1, c2 = (Tt + 7) // 8 * 8, (Ti + 7) // 8 * 8
    q = torch.randn(B * Nq, D, generator=g).to(BF).cuda()
    k1 = torch.randn(B * Tt, D, generator=g).to(BF).cuda()
    k2 = torch.randn(B * Ti, D, generator=g).to(BF).cuda()
    v1t = torch.zeros(D, (B - 1) * c1 + (Tt + 63) // 64 * 64, dtype=BF, device="cuda")
    v2t = torch.zeros(D, (B - 1) * c2 + (Ti + 63) // 64 * 64, dtype=BF, device="cuda")
    for b in range(B):
        v1t[:, b * c1: b * c1 + Tt] = torch.randn(D, Tt, generator=g).to(BF).cuda()
        v2t[:, b * c2: b * c2 + Ti] = torch.randn(D, Ti, generator=g).to(BF).cuda()
    o = ops.attention_2seg_vt(q, k1, v1t, Tt, k2, v2t, Ti, H, batch=B, cols1=c1, cols2=c2)
    q_ref, s_ref = ops.quant_rows_mxfp8(o)
    q8 = torch.zeros((B * Nq, D), dtype=torch.uint8, device="cuda")
    s8 = torch.zeros((ops.mx_scale_bytes(B * Nq, D),), dtype=torch.uint8, device="cuda")
    ops.attention_2seg_vt(q, k1, v1t, Tt, k2, v2t, Ti, H, batch=B, cols1=c1, cols2=c2, out8=q8, scale8=s8)
    assert torch.equal(q8, q_ref), (q8.view(torch.float8_e4m3fn).float() - q_ref.view(torch.float8_e4m3fn).float()).abs().max()
    assert torch.equal(ops.mx_scales_to_rows(s8, B * Nq, D), ops.mx_scales_to_rows(s_ref, B * Nq, D))


def test_dit_forward_mx_mode_fused_quantisers_equal_the_unfused_form():
    """fp8 mode with MX block scales: the forward with every fused quantiser (LN, FFN-up epilogue, both attention epilogues) is
    bit-identical to the forward that quantises in separate passes (`fp8_fuse_quant = False`)."""
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    from oracle import dit_oracle as O
    cfg = O.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    p = O.make_synthetic_params(cfg, dtype=BF)
    lat, text, image = O.make_synthetic_inputs(cfg, 2, 16, 24, dtype=BF, text_len=40, real_text=8)
    outs = []
    for fuse in (True, False):
        m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64,
                                         added_kv_proj_dim=256, device="cuda:0")
        m.load_synthetic_({k: v.cuda() for k, v in p.items()})
        m.fp8_fuse_quant = fuse
        m.enable_fp8_gemms(mx=True).enable_fp8_attention()
        outs.append(m(lat.cuda(), torch.tensor([500], device="cuda:0"), text.cuda(), image.cuda()).sample.float().cpu())
        del m
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max()


def test_gemm_mxfp8_rejects_bad_shapes():
    from chronoedit_amd import ops
    aq = torch.zeros(8, 128, dtype=torch.uint8, device="cuda")
    wq = torch.zeros(16, 128, dtype=torch.uint8, device="cuda")
    sa = torch.ones(512, dtype=torch.uint8, device="cuda")
    sw = torch.ones(512, dtype=torch.uint8, device="cuda")
    with pytest.raises(ops.HipKernelError):
        ops.gemm_mxfp8(aq, sa, wq, sw, None)  # K % 256 != 0
