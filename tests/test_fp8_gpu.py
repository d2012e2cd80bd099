"""fp8 (OCP e4m3) GEMM path (BASELINE.json configs[4]): the row quantiser against torch's float8_e4m3fn cast, the MX-instruction
GEMM against fp32 math on the dequantised operands (what the kernel is defined to compute), and the end-to-end quantisation
error of an fp8 Linear against the bf16 one (reported)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _deq(q, s):
    return q.cpu().view(torch.float8_e4m3fn).float() * s.cpu()[:, None]


@pytest.mark.parametrize("M,K", [(7, 256), (300, 5120), (1000, 13824)])
def test_quant_rows_fp8_matches_torch_cast(M, K):
    from chronoedit_amd import ops
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3).to(BF)
    x[min(2, M - 1)] = 0  # an all-zero row
    q, s = ops.quant_rows_fp8(x.cuda())
    amax = x.float().abs().amax(dim=1)
    want_s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.allclose(s.cpu(), want_s, rtol=1e-6, atol=0)
    want_q = (x.float() / want_s[:, None]).to(torch.float8_e4m3fn)
    got = q.cpu().view(torch.float8_e4m3fn)
    same = (got.view(torch.uint8) == want_q.view(torch.uint8)).float().mean().item()
    assert same > 0.999, same  # RNE on both sides; the division by the scale may differ in the last fp32 bit at exact ties
    assert (got.float() - want_q.float()).abs().max().item() <= 32.0  # never more than one fp8 step (32 at the top binade)
    assert got.float().abs().max().item() <= 448.0
    assert rel_l2(_deq(q, s), x) < 4e-2  # e4m3: 3 mantissa bits


@pytest.fixture(params=[0, 1], ids=["w8", "w4"])
def fp8_variant(request):
    """Both main loops of ce_gemm_fp8: 8 waves / 4 phases (ce_gemm_fp8.hip) and one wave per SIMD (ce_gemm_fp8w4.hip)."""
    from chronoedit_amd import ops
    old = ops.set_gemm_fp8_variant(request.param)
    yield request.param
    ops.set_gemm_fp8_variant(old)


@pytest.mark.parametrize("M,N,K,epi", [(256, 256, 256, "bias"), (300, 520, 512, "bias"), (1000, 1280, 5120, "gelu"),
                                       (7200, 5120, 13824, "gate"), (14400, 15360, 5120, "bias"), (13068, 5120, 5120, "gate")])
def test_gemm_fp8_matches_fp32_on_dequantised_operands(M, N, K, epi, fp8_variant):
    from chronoedit_amd import ops
    g = torch.Generator().manual_seed(2)
    a = torch.randn(M, K, generator=g).to(BF).cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).to(BF).cuda()
    w[: min(N, 5)] *= 4  # asymmetric rows: a transposed operand would show
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(BF).cuda()
    gate = torch.randn(N, generator=g).cuda()
    aq, sa = ops.quant_rows_fp8(a)
    wq, sw = ops.quant_rows_fp8(w)
    kw = {"bias": dict(), "gelu": dict(epilogue=ops.EPI_BIAS_GELU), "gate": dict(epilogue=ops.EPI_GATE_RES, gate=gate, res=res)}[epi]
    out = ops.gemm_fp8(aq, sa, wq, sw, bias, **kw)
    # what the kernel is defined to compute, in fp32 on the GPU (large shapes) from the dequantised fp8 operands
    ad = aq.view(torch.float8_e4m3fn).float() * sa[:, None]
    wd = wq.view(torch.float8_e4m3fn).float() * sw[:, None]
    lin = ad @ wd.t() + bias
    ref = {"bias": lin, "gelu": torch.nn.functional.gelu(lin.to(BF).float(), approximate="tanh"),
           "gate": res.float() + lin.to(BF).float() * gate}[epi]
    e = rel_l2(out, ref)
    assert e < 4e-3, e  # bf16 rounding of the output only
    # and the price of fp8 itself against the bf16 Linear (informational bound)
    full = a.float() @ w.float().t() + bias
    e8 = rel_l2(lin, full)
    print(f"fp8 GEMM {M}x{N}x{K} {epi}: kernel-vs-definition {e:.2e}; fp8 quantisation error vs bf16 operands {e8:.2e}")
    assert e8 < 6e-2


def test_gemm_fp8_rejects_bad_shapes():
    from chronoedit_amd import ops
    aq = torch.zeros(8, 128, dtype=torch.uint8, device="cuda")
    wq = torch.zeros(16, 128, dtype=torch.uint8, device="cuda")
    s = torch.ones(8, device="cuda")
    with pytest.raises(ops.HipKernelError):
        ops.gemm_fp8(aq, s, wq, torch.ones(16, device="cuda"), None)  # K % 256 != 0


@pytest.mark.parametrize("mx", [False, True], ids=["row-scales", "mx-block-scales"])
def test_dit_forward_fp8_mode_vs_fp8_contract_oracle(mx):
    """enable_fp8_gemms(mx=...): the HIP forward against the fp32 oracle that restates the same fp8 contract (fake-quantised operands of
    the six large Linears per block; per-row scales or OCP-MX block scales), and - reported - against the un-quantised fp32 oracle
    (the price of fp8 itself)."""
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    from oracle import dit_oracle as O
    cfg = O.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=4, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    p_bf = O.make_synthetic_params(cfg, seed=3, dtype=BF)
    lat, text, image = O.make_synthetic_inputs(cfg, 2, 16, 24, dtype=BF)
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=4, text_dim=128, image_dim=64,
                                     added_kv_proj_dim=256, device="cuda:0")
    m.load_synthetic_({k: v.cuda() for k, v in p_bf.items()})
    ts = torch.tensor([400], device="cuda:0")
    args = (lat.cuda(), ts, text.cuda(), image.cuda())
    out16 = m(*args, return_dict=False)[0].clone()
    m.enable_fp8_gemms(mx=mx)
    out8 = m(*args, return_dict=False)[0].clone()
    m.enable_fp8_gemms(False)
    assert torch.equal(m(*args, return_dict=False)[0], out16)  # switching back restores the bf16 path bit for bit
    p32 = {k: v.float() for k, v in p_bf.items()}
    with torch.no_grad():
        ref = O.dit_forward(p32, cfg, lat.float(), torch.tensor([400]), text.float(), image.float())
        ref8 = O.dit_forward(p32, cfg, lat.float(), torch.tensor([400]), text.float(), image.float(), fp8="mx" if mx else True)
    e_contract, e_price, e_bf16 = rel_l2(out8, ref8), rel_l2(ref8, ref), rel_l2(out16, ref)
    print(f"fp8 mode ({'MX block' if mx else 'per-row'} scales): hip-vs-fp8-contract-oracle {e_contract:.3e}; fp8 contract vs exact fp32 {e_price:.3e}; bf16 path vs fp32 {e_bf16:.3e}")
    assert e_contract <= 2.5e-2      # bf16-level agreement with the contract (rounding boundaries of fp8 add a little)
    assert rel_l2(out8, ref) <= 0.12  # the whole fp8 forward stays close to the exact one
    assert rel_l2(out8, out16) > 1e-3


@pytest.mark.parametrize("linears", [("qkv",), ("o1",), ("q2",), ("o2",), ("f1",), ("f2",), ("f1", "f2"), "accurate", ("qkv", "o2", "f2")],
                         ids=lambda v: v if isinstance(v, str) else "+".join(v))
def test_dit_forward_mixed_precision_vs_contract_oracle(linears):
    """Round 6 - enable_fp8_gemms(linears= / policy=): any subset of the six large Linears on the MX fp8 GEMM, the rest on the bf16 GEMM (every
    producer / consumer pairing: LayerNorm -> fp8 or bf16, attention -> fused MX operand or bf16, the FFN pair fused / split), with the MXFP8
    self-attention on.  Against the oracle restating the SAME subset (dit_forward(fp8="mx", fp8_linears=...)): contract-level agreement, and the
    selection really is per Linear (a subset differs from both the bf16 path and the all-six mode)."""
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    from oracle import dit_oracle as O
    cfg = O.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=3, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    p_bf = O.make_synthetic_params(cfg, seed=5, dtype=BF)
    lat, text, image = O.make_synthetic_inputs(cfg, 2, 16, 24, dtype=BF)
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=3, text_dim=128, image_dim=64,
                                     added_kv_proj_dim=256, device="cuda:0")
    m.load_synthetic_({k: v.cuda() for k, v in p_bf.items()})
    ts = torch.tensor([400], device="cuda:0")
    args = (lat.cuda(), ts, text.cuda(), image.cuda())
    out16 = m(*args, return_dict=False)[0].clone()
    m.enable_fp8_gemms().enable_fp8_attention()
    out_all = m(*args, return_dict=False)[0].clone()
    if isinstance(linears, str):
        m.enable_fp8_gemms(policy=linears)
        names = m.FP8_POLICIES[linears]
    else:
        m.enable_fp8_gemms(linears=linears)
        names = linears
    assert m.fp8_linears == tuple(n for n in m.FP8_LINEARS if n in names)
    out = m(*args, return_dict=False)[0].clone()
    again = m(*args, return_dict=False)[0]
    assert torch.equal(out, again) and torch.isfinite(out.float()).all()
    p32 = {k: v.float() for k, v in p_bf.items()}
    with torch.no_grad():
        ref = O.dit_forward(p32, cfg, lat.float(), torch.tensor([400]), text.float(), image.float(), fp8="mx", fp8_attn=True, fp8_linears=names)
    e = rel_l2(out, ref)
    print(f"mixed precision {names}: hip vs the same-subset contract oracle {e:.3e}; vs bf16 path {rel_l2(out, out16):.3e}; vs all six {rel_l2(out, out_all):.3e}")
    assert e <= 2.5e-2
    assert not torch.equal(out, out16) and not torch.equal(out, out_all)
    with pytest.raises(ValueError):
        m.enable_fp8_gemms(linears=("qkv", "nope"))
    with pytest.raises(ValueError):
        m.enable_fp8_gemms(policy="fast", linears=("qkv",))


def test_ln_affine_fp8_equals_two_launch_form():
    """The fused LayerNorm -> fp8 kernel == ln_affine followed by quant_rows_fp8, bit for bit (bytes and scales)."""
    from chronoedit_amd import ops
    g = torch.Generator().manual_seed(4)
    M, D = 1003, 5120
    x = (torch.randn(M, D, generator=g) * 2 + 0.3).to(BF).cuda()
    a = (1 + 0.2 * torch.randn(2, D, generator=g)).cuda()
    b = (0.1 * torch.randn(2, D, generator=g)).cuda()
    for kw in (dict(), dict(ab_rows=600, ab_stride=D)):
        aa, bb = (a, b) if kw else (a[0], b[0])
        h = ops.ln_affine(x, aa, bb, 1e-6, **kw)
        q_ref, s_ref = ops.quant_rows_fp8(h)
        q = torch.empty((M, D), dtype=torch.uint8, device="cuda")
        s = torch.empty((M,), dtype=torch.float32, device="cuda")
        ops.ln_affine_fp8(x, aa, bb, 1e-6, out=q, scale=s, **kw)
        assert torch.equal(s, s_ref) and torch.equal(q, q_ref)


@pytest.mark.parametrize("mx", [False, True], ids=["row-scales", "mx-block-scales"])
def test_edit_end_to_end_fp8_mode_vs_fp8_contract_oracle(mx):
    """A whole 4-step CFG edit (prepare_latents -> loop -> decode) with the transformer in fp8 GEMM mode, against the oracle edit
    whose DiT follows the same fp8 contract; and how far that is from the exact-arithmetic edit (reported)."""
    from chronoedit_amd.pipeline import ChronoEditPipeline
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    from chronoedit_amd.vae import AutoencoderKLWan
    from oracle import dit_oracle as D
    from oracle import pipeline_oracle as P
    from oracle import vae_oracle as V
    dcfg = D.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    vcfg = V.VAEConfig(dim=32, z_dim=16)
    dp = D.make_synthetic_params(dcfg, dtype=BF)
    vp = V.make_synthetic_params(vcfg)
    g = torch.Generator().manual_seed(0)
    H, W, F = 64, 96, 5
    image = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    prompt, negative = torch.randn(1, 40, 128, generator=g), torch.randn(1, 40, 128, generator=g)
    img_emb = torch.randn(1, 257, 64, generator=g)
    lat0 = torch.randn(1, 16, 2, H // 8, W // 8, generator=g)
    dp32 = {k: v.float() for k, v in dp.items()}
    cpu_args = (image.to(BF).float(), prompt.to(BF).float(), negative.to(BF).float(), img_emb.to(BF).float())
    with torch.no_grad():
        lat8, vid8 = P.edit(dp32, dcfg, vp, vcfg, *cpu_args, lat0.clone(), num_frames=F, steps=4, fp8="mx" if mx else True)
        lat_x, _ = P.edit(dp32, dcfg, vp, vcfg, *cpu_args, lat0.clone(), num_frames=F, steps=4, decode=False)
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64,
                                     added_kv_proj_dim=256, device="cuda:0")
    m.load_synthetic_({k: v.cuda() for k, v in dp.items()})
    m.enable_fp8_gemms(mx=mx)
    pipe = ChronoEditPipeline(vae=AutoencoderKLWan({k: v.cuda() for k, v in vp.items()}, dim=32, z_dim=16), transformer=m,
                              scheduler=FlowUniPCMultistepScheduler(flow_shift=5.0))
    args = (image.cuda().to(BF), prompt.cuda().to(BF), negative.cuda().to(BF), img_emb.cuda().to(BF))
    lat = pipe.edit_tensors(*args, num_frames=F, num_inference_steps=4, guidance_scale=5.0, latents=lat0.cuda(), output_type="latent")
    vid = pipe.edit_tensors(*args, num_frames=F, num_inference_steps=4, guidance_scale=5.0, latents=lat0.cuda())
    print(f"fp8 edit: latents vs fp8-contract oracle {rel_l2(lat, lat8):.3e}, video {rel_l2(vid, vid8):.3e}; "
          f"contract vs exact edit (latents) {rel_l2(lat8, lat_x):.3e}")
    # measured on MI355X (round 6): 1.6e-2 / 2.2e-2 against the contract oracle, the contract itself 1.85e-2 from the exact edit at this toy width;
    # the statement of what fp8 costs where it matters - the full width, both policies - is tests/test_width_depth_gpu.py (the 0.2 that
    # stood here said nothing)
    assert rel_l2(lat, lat8) < 4e-2 and rel_l2(vid, vid8) < 5e-2
    assert rel_l2(lat, lat_x) < 6e-2


# ---------------------------------------------------------------------------------------------------------------------
# MXFP8 self-attention ("fp8 weights+attn" of BASELINE.json configs[4]; contract: oracle.dit_oracle.attention_mxfp8)
# ---------------------------------------------------------------------------------------------------------------------
def _dequant(q8: torch.Tensor, s8: torch.Tensor, axis_blocks_last: bool = True) -> torch.Tensor:
    """e4m3 bytes [.., n] + E8M0 bytes [.., n/32] -> fp32."""
    x = q8.view(torch.float8_e4m3fn).float()
    sc = torch.exp2(s8.float() - 127.0).repeat_interleave(32, dim=-1)
    return x * sc


@pytest.mark.parametrize("M,H,rope", [(70, 2, True), (130, 40, True), (64, 4, False)])
def test_mxfp8_qk_producer_matches_contract(M, H, rope):
    """ce_rmsnorm_rope_mxfp8 == MX-quantise(what ce_rmsnorm_rope_bf16 stores), bit for bit (RNE on both sides)."""
    from chronoedit_amd import ops
    from oracle import dit_oracle as O
    D = H * 128
    g = torch.Generator().manual_seed(31)
    buf = torch.randn(M, 3 * D, generator=g).to(torch.bfloat16).cuda()
    w = (1 + 0.05 * torch.randn(D, generator=g)).cuda()
    cs = None
    if rope:
        ang = torch.rand(M, 64, generator=g, dtype=torch.float64) * 6.28
        cs = torch.stack([ang.cos(), ang.sin()], -1).float().cuda()
    q8, s8 = ops.rmsnorm_rope_mxfp8(buf[:, D:2 * D], w, cs, 128, 1e-6)
    ref = buf.clone()
    ops.rmsnorm_rope_(ref[:, D:2 * D], w, cs, 128, 1e-6)
    want = O.mx_quant(ref[:, D:2 * D].float().cpu(), -1)
    got = _dequant(q8.cpu(), s8.cpu())
    assert torch.equal(got, want), (got - want).abs().max()
    assert torch.equal(buf, torch.randn(M, 3 * D, generator=torch.Generator().manual_seed(31)).to(torch.bfloat16).cuda())  # source untouched


@pytest.mark.parametrize("N,H,B", [(64, 2, 1), (200, 4, 2), (7200 // 8, 2, 1)])
def test_mxfp8_v_transpose_matches_contract(N, H, B):
    """ce_v_mxfp8_transpose: de-quantised and un-permuted it equals MX-quantise(V) over blocks of 32 consecutive keys; the key
    order inside a 64-key tile is the accumulator-register order of the attention kernel; padded keys are zero."""
    from chronoedit_amd import ops
    from oracle import dit_oracle as O
    D = H * 128
    g = torch.Generator().manual_seed(32)
    buf = (torch.randn(B * N, 3 * D, generator=g) * 1.7).to(torch.bfloat16).cuda()
    v8t, sv = ops.v_mxfp8_transpose(buf[:, 2 * D:], N, B, H)
    npad = v8t.shape[-1]
    assert npad == (N + 63) // 64 * 64 and sv.shape == (B, H, npad // 64, 128, 2)
    pos = torch.arange(npad)
    tile, p = pos // 64, pos % 64
    gsel, j = p // 32, p % 32
    key = tile * 64 + 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * gsel  # key stored at position pos
    vt = torch.zeros(B, H, 128, npad)
    vt[..., key] = v8t.cpu().view(torch.float8_e4m3fn).float()                   # elements back in key order
    svk = sv.cpu().permute(0, 1, 3, 2, 4).reshape(B, H, 128, npad // 32)         # [.., d, 32-key block]
    vt = vt * torch.exp2(svk.float() - 127.0).repeat_interleave(32, dim=-1)       # scale of 32 consecutive keys
    v = buf[:, 2 * D:].float().cpu().view(B, N, H, 128).permute(0, 2, 1, 3)  # [B, H, N, 128]
    vp = torch.zeros(B, H, npad, 128)
    vp[:, :, :N] = v
    want = O.mx_quant(vp, 2).permute(0, 1, 3, 2)  # blocks of 32 consecutive keys
    assert torch.equal(vt, want), (vt - want).abs().max()


@pytest.mark.parametrize("N,H,B,spread", [(64, 2, 1, 1.0), (200, 2, 1, 1.0), (333, 4, 2, 1.0), (1000, 2, 1, 3.0), (700, 2, 1, 40.0),
                                          (300, 8, 2, 1.0), (560, 16, 1, 2.0)])  # H % 8 == 0: the XCD-aware work order, remainder blocks last
def test_mxfp8_attention_kernel_vs_contract(N, H, B, spread):
    """The three kernels together vs oracle.attention_mxfp8 (same quantisation, same online order, same offset schedule - lazy for
    the default kernel, per tile for the plain loop): rel-L2 <= 1.5e-2 (what differs is the fp32 summation order and exp2 at the
    last ulp, which can flip individual e4m3 roundings of P or, at spread 40, a knife-edge offset decision); the contract's own
    distance from exact fp32 attention is printed beside it.  spread 40 forces the exact route of the default kernel (scores
    climbing past the speculative window tile after tile, through exp2 overflow)."""
    from chronoedit_amd import ops
    from oracle import dit_oracle as O
    D = H * 128
    g = torch.Generator().manual_seed(33)
    qkv = torch.randn(B * N, 3 * D, generator=g)
    qkv[:, :2 * D] *= spread ** 0.5  # spread > 1: peakier softmax (forces running-max updates and rescales)
    qkv = qkv.to(torch.bfloat16)
    one = torch.ones(D).cuda()
    dev = qkv.cuda()
    q8, sq = ops.rmsnorm_rope_mxfp8(dev[:, :D], one, None, 128, 1e-6, post_scale=ops.MXFP8_Q_SCALE)
    k8, sk = ops.rmsnorm_rope_mxfp8(dev[:, D:2 * D], one, None, 128, 1e-6)
    v8t, sv = ops.v_mxfp8_transpose(dev[:, 2 * D:], N, B, H)
    outs = {}
    for variant in (0, 1):  # plain loop / software-pipelined (default)
        ops.set_attention_mxfp8_variant(variant)
        outs[variant] = ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, batch=B).float().cpu()
    ops.set_attention_mxfp8_variant(1)
    out = outs[1]
    # the oracle on the same normalised q / k (weight 1, no rope)
    ref_in = dev.clone()
    ops.rmsnorm_rope_(ref_in[:, :D], one, None, 128, 1e-6, x2=ref_in[:, D:2 * D], w2=one)
    f = lambda t: t.float().cpu().view(B, N, H, 128).permute(0, 2, 1, 3)
    q, k, v = f(ref_in[:, :D]), f(ref_in[:, D:2 * D]), f(ref_in[:, 2 * D:])
    want = O.attention_mxfp8(q, k, v).permute(0, 2, 1, 3).reshape(B * N, D)                      # default kernel: lazy offset
    want0 = O.attention_mxfp8(q, k, v, lazy_offset=False).permute(0, 2, 1, 3).reshape(B * N, D)  # plain loop: offset moves every tile
    exact = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(B * N, D)
    e_k, e_k0, e_c = rel_l2(out, want), rel_l2(outs[0], want0), rel_l2(want, exact)
    print(f"mxfp8 attention N={N} H={H} B={B} spread={spread}: kernel vs contract {e_k:.3e} (plain loop {e_k0:.3e}); contract vs exact fp32 {e_c:.3e}")
    assert torch.isfinite(out).all() and e_k < 1.5e-2 and e_k0 < 1.5e-2


@pytest.mark.parametrize("Nq,Tt,Ti,H,B", [(200, 40, 257, 2, 1), (333, 512, 257, 8, 2), (96, 64, 65, 2, 2)])
def test_mxfp8_two_segment_attention_vs_contract(Nq, Tt, Ti, H, B):
    """The cross-attention under the MXFP8 contract (round 5): out = bf16(attention_mxfp8(q, k_text, v_text)) + bf16(attention_mxfp8(q,
    k_image, v_image)) as two launches - the text segment plain, the image segment through ce_attention_mxfp8_add (also in place, and as
    the out-projection's MX operand: bit-identical to quant_rows_mxfp8 of the bf16 sum).  vs the oracle's two calls: rel-L2 <= 1.5e-2."""
    from chronoedit_amd import ops
    from oracle import dit_oracle as O
    D = H * 128
    g = torch.Generator().manual_seed(35)
    one = torch.ones(D).cuda()
    qd = torch.randn(B * Nq, D, generator=g).to(torch.bfloat16).cuda()
    segs = []
    for n in (Tt, Ti):
        kv = torch.randn(B * n, 2 * D, generator=g).to(torch.bfloat16).cuda()
        k8, sk = ops.rmsnorm_rope_mxfp8(kv[:, :D], one, None, 128, 1e-6)
        v8t, sv = ops.v_mxfp8_transpose(kv[:, D:], n, B, H)
        segs.append((kv, (k8, sk, v8t, sv)))
    q8, sq = ops.rmsnorm_rope_mxfp8(qd, one, None, 128, 1e-6, post_scale=ops.MXFP8_Q_SCALE)
    o_t = ops.attention_mxfp8(q8, sq, *segs[0][1], H, batch=B)
    o_sum = ops.attention_mxfp8(q8, sq, *segs[1][1], H, batch=B, out=torch.empty_like(o_t), add=o_t)
    o_inplace = o_t.clone()
    ops.attention_mxfp8(q8, sq, *segs[1][1], H, batch=B, out=o_inplace, add=o_inplace)
    assert torch.equal(o_inplace, o_sum)
    o_i = ops.attention_mxfp8(q8, sq, *segs[1][1], H, batch=B)
    assert torch.equal(o_sum, (o_t.float() + o_i.float()).to(torch.bfloat16))  # each segment rounded to bf16, then a bf16 add
    # the fused MX operand of the out-projection == the row quantiser on the bf16 sum
    o8 = torch.empty(B * Nq, D, dtype=torch.uint8, device="cuda")
    s8 = torch.zeros(ops.mx_scale_bytes(B * Nq, D), dtype=torch.uint8, device="cuda")
    ops.attention_mxfp8(q8, sq, *segs[1][1], H, batch=B, out8=o8, scale8=s8, add=o_t)
    q_ref, s_ref = ops.quant_rows_mxfp8(o_sum)
    assert torch.equal(o8, q_ref)
    assert torch.equal(ops.mx_scales_to_rows(s8, B * Nq, D), ops.mx_scales_to_rows(s_ref, B * Nq, D))
    # the oracle on the same normalised operands
    qn = qd.clone()
    ops.rmsnorm_rope_(qn, one, None, 128, 1e-6)
    f = lambda t, n: t.float().cpu().view(B, n, H, 128).permute(0, 2, 1, 3)
    want = None
    for (kv, _), n in zip(segs, (Tt, Ti)):
        kn = kv[:, :D].clone()
        ops.rmsnorm_rope_(kn, one, None, 128, 1e-6)
        w = O.attention_mxfp8(f(qn, Nq), f(kn, n), f(kv[:, D:], n)).permute(0, 2, 1, 3).reshape(B * Nq, D).to(torch.bfloat16)
        want = w if want is None else (want.float() + w.float()).to(torch.bfloat16)
    e = rel_l2(o_sum.float().cpu(), want.float())
    print(f"two-segment MXFP8 attention Nq={Nq} Tt={Tt} Ti={Ti} H={H} B={B}: kernel vs contract {e:.3e}")
    assert torch.isfinite(o_sum.float()).all() and e < 1.5e-2


@pytest.mark.parametrize("mx,cross", [(False, False), (True, False), (True, True)], ids=["row-scales", "mx-block-scales", "mx-block-scales+fp8-cross-attention"])
def test_dit_forward_with_mxfp8_attention_vs_contract_oracle(mx, cross):
    """fp8 GEMMs + MXFP8 self-attention (bench.py --fp8) through the whole DiT: vs the oracle restating both contracts (<= 2.5e-2), and
    the mode's distance from exact fp32 bounded against the bf16 path's as SURVEY section 8c prescribes (<= 10 x the bf16 error)."""
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    from oracle import dit_oracle as O
    cfg = O.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=4, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    p = O.make_synthetic_params(cfg, dtype=torch.bfloat16)
    lat, text, image = O.make_synthetic_inputs(cfg, 2, 16, 24, dtype=torch.bfloat16, text_len=40, real_text=8)
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=4, text_dim=128, image_dim=64,
                                     added_kv_proj_dim=256, device="cuda:0")
    m.load_synthetic_({k: v.cuda() for k, v in p.items()})
    ts = torch.tensor([500], device="cuda:0")
    args = (lat.cuda(), ts, text.cuda(), image.cuda())
    out_bf16 = m(*args).sample.float().cpu()
    m.enable_fp8_gemms(mx=mx).enable_fp8_attention(cross=cross)  # cross=True: the opt-in MXFP8 cross-attention (round 5; off by default)
    out_fp8 = m(*args).sample.float().cpu()
    m.enable_fp8_gemms(False)
    out_attn_only = m(*args).sample.float().cpu()
    pf = {k: v.float() for k, v in p.items()}
    with torch.no_grad():
        a32 = (lat.float(), torch.tensor([500]), text.float(), image.float())
        exact = O.dit_forward(pf, cfg, *a32)
        contract = O.dit_forward(pf, cfg, *a32, fp8="mx" if mx else True, fp8_attn="all" if cross else True)
    e_bf16, e_fp8, e_attn = rel_l2(out_bf16, exact), rel_l2(out_fp8, exact), rel_l2(out_attn_only, exact)
    e_contract = rel_l2(out_fp8, contract)
    print(f"DiT 4 blocks: bf16 path vs exact {e_bf16:.3e} | fp8 GEMMs + MXFP8 attention vs exact {e_fp8:.3e} (attention only: {e_attn:.3e}) | "
          f"vs the fp8 contract oracle {e_contract:.3e}")
    assert e_contract < 2.5e-2 and e_fp8 < 10 * e_bf16 + 1e-2
