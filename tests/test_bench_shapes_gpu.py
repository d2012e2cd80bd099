"""Parity at the shapes bench.py REPORTS but the smaller tests do not reach ("untested configs at real width"):
  * BASELINE.json configs[3]: 28 800 keys (450 key tiles of online softmax) in the plain layout and in the blocked
    [source rank][sample][local token] receive layout at the 8-GPU per-rank shape (5 heads, shards rounded up to 3 648 rows);
  * configs[4]: one full-width block at 1584x1056 (N = 13 068: not a multiple of 64 / 256 - remainder paths of the attention
    kernel and of the 256-tile GEMM incl. its split-K tail), bf16 and fp8 mode;
  * fp8 mode at full width (D = 5120, K = 13 824 rows under one e4m3 scale) at N = 7 200, and at the model's depth (40 blocks,
    narrow width), bounded against the bf16 path's own error as SURVEY section 8c prescribes (e_fp8 <= k x e_bf16, k stated per test:
    measured 8.0 for ONE full-width block, 2.6 after 40 blocks; e4m3 carries 3 mantissa bits - 3.6 % rms per operand - so a
    product of two quantised operands is ~10 x noisier than its bf16 form whatever the scale granularity).
References: fp32 torch SDPA on the device for the attention kernels (the operands are bf16-exact), the fp32 CPU oracle
(oracle/dit_oracle.py, restating transformer_chronoedit.py:38-476) for the blocks."""
import pytest
import torch

from oracle import device as OD
from oracle import dit_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _sdpa_rows(q, k, v, H, rows):
    """fp32 softmax(q k^T / sqrt(128)) v for the query rows `rows` (a slice), per head; q [Nq, H*128], k / v [Nkv, H*128]."""
    qh = q[rows].float().view(-1, H, 128).transpose(0, 1)
    kh = k.float().view(-1, H, 128).transpose(0, 1)
    vh = v.float().view(-1, H, 128).transpose(0, 1)
    s = torch.softmax(qh @ kh.transpose(1, 2) * 128 ** -0.5, dim=-1)
    return (s @ vh).transpose(0, 1).reshape(-1, H * 128)


def test_attention_at_28800_keys_plain_layout():
    """configs[3] on one GPU: every query row sees 28 800 keys = 450 key tiles (the 7 200-key tests stop at 113)."""
    from chronoedit_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(41)
    N, H = 28800, 2
    D = H * 128
    qkv = torch.randn(N, 3 * D, generator=g).to(BF).to(dev)
    qkv[:, 2 * D:].mul_(torch.linspace(0.5, 1.5, D, device=dev).to(BF))
    qkv[:, 2 * D:].add_((torch.arange(N, device=dev) % 7).to(BF)[:, None] * 0.25)  # key-dependent v: a permuted P.V shows up
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    vt = ops.v_transpose(v, H)
    out = ops.attention_vt(q, k, vt, H)
    base = ops.attention(q, k, v, H)
    assert rel_l2(out, base) < 3e-3
    worst = 0.0
    for r0 in range(0, N, 4800):  # the whole tensor, in slabs (scores in fp32: 4800 x 28800 x 2 heads = 1.1 GB)
        rows = slice(r0, r0 + 4800)
        e = rel_l2(out[rows], _sdpa_rows(q, k, v, H, rows))
        worst = max(worst, e)
    print(f"attention 28800 x 28800, plain layout: worst slab rel-L2 vs fp32 {worst:.3e}")
    assert worst < 1e-2


def test_attention_at_28800_keys_blocked_layout_at_the_8_gpu_rank_shape():
    """One rank of the 8-way Ulysses split of configs[3] with the guidance pair batched: 5 heads, 2 samples, rows in the all-to-all
    receive layout [8 source ranks][2 samples][3648 local tokens] (3600 rounded up to a multiple of 64; 28 800 valid tokens) ==
    the plain-layout kernel on the un-blocked tensors bit for bit, and <= 1e-2 from fp32 attention."""
    from chronoedit_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(43)
    N, W, H, B = 28800, 8, 5, 2
    n = (N // W + 63) // 64 * 64
    D, T = H * 128, W * n
    plain = torch.randn(B, T, 3 * D, generator=g).to(BF).to(dev)  # [sample][token][q | k | v]
    plain[:, N:, D:] = 0.5                                          # padded keys: finite junk the mask must hide
    blocked = plain.view(B, W, n, 3 * D).permute(1, 0, 2, 3).contiguous().view(W * B * n, 3 * D)
    vt = ops.v_transpose_blocked(blocked[:, 2 * D:], H, B, n, N)
    out = ops.attention_vt_blocked(blocked[:, :D], blocked[:, D:2 * D], vt, H, B, n, N)
    got = out.view(W, B, n, D).permute(1, 0, 2, 3).reshape(B, T, D)
    for b in range(B):
        vt_b = ops.v_transpose(plain[b, :N, 2 * D:], H)
        want = ops.attention_vt(plain[b, :, :D], plain[b, :N, D:2 * D], vt_b, H)
        assert torch.equal(got[b], want), (b, (got[b].float() - want.float()).abs().max())
        for rows in (slice(0, 2400), slice(14000, 16400), slice(26400, 28800)):  # first / middle / last source blocks
            e = rel_l2(got[b][rows], _sdpa_rows(plain[b, :N, :D], plain[b, :N, D:2 * D], plain[b, :N, 2 * D:], H, rows))
            assert e < 1e-2, (b, rows, e)


def test_attention_at_the_literal_121_frame_stress_shape():
    """BASELINE.json's "121-frame reasoning mode" read literally (optional stress shape - NOT reference
    behaviour): 121 pixel frames -> 31 latent frames at 720p = 111 600 tokens, 1 744 key tiles of online softmax per query row.  Two
    heads, every 12th 2 400-row slab against fp32 softmax on the device; the V^T operand's 32-bit DMA offsets (111 600 keys x 256 B rows)
    and the work order at 436 query blocks per head."""
    from chronoedit_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(47)
    N, H = 111600, 2
    D = H * 128
    qkv = torch.randn(N, 3 * D, generator=g).to(BF).to(dev)
    qkv[:, 2 * D:].add_((torch.arange(N, device=dev) % 7).to(BF)[:, None] * 0.25)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    out = ops.attention_vt(q, k, ops.v_transpose(v, H), H)
    assert torch.isfinite(out.float()).all()
    worst = 0.0
    for r0 in list(range(0, N - 2400, 12 * 2400)) + [N - 2400]:  # (scores in fp32: 2400 x 111600 x 2 heads = 2.1 GB per slab)
        rows = slice(r0, r0 + 2400)
        worst = max(worst, rel_l2(out[rows], _sdpa_rows(q, k, v, H, rows)))
    print(f"attention 111600 x 111600 (31 latent frames at 720p): worst sampled slab rel-L2 vs fp32 {worst:.3e}")
    assert worst < 1e-2


def test_full_width_block_at_the_121_frame_stress_shape():
    """One block of the 14B width at [1, 36, 31, 90, 160] with rope_temporal_skip_len = 31 (the constructor parameter the stress reading
    needs: transformer_chronoedit.py:205-209 asserts num_frames in {2, skip_len}): N = 111 600 token rows through every kernel of the
    step - 436-tile GEMM M axes, 1 744 key tiles per attention row, ~10 GB of workspaces.  No CPU oracle finishes this size (attention
    couples all frames, so no crop is comparable): the statement here is finite + deterministic, the temporal RoPE table at 31 plain
    positions against the fp64 oracle table, and the per-kernel parity at this key count by the attention test above and the GEMM tests."""
    cfg = O.DiTConfig(num_layers=1, rope_temporal_skip_len=31)
    p_bf = O.make_synthetic_params(cfg, seed=7, dtype=BF)
    lat, text, image = O.make_synthetic_inputs(cfg, 31, 90, 160, dtype=BF)
    model = _build(cfg, p_bf)
    ts = torch.tensor([800], device="cuda:0")
    args = (lat.cuda(), ts, text.cuda(), image.cuda())
    out = model(*args, return_dict=False)[0]
    again = model(*args, return_dict=False)[0]
    assert out.shape == (1, 16, 31, 90, 160) and torch.isfinite(out.float()).all()
    assert torch.equal(out, again)
    # the temporal RoPE table at 31 frames against the fp64 oracle table
    from chronoedit_amd.transformer import rope_cos_sin
    cs = rope_cos_sin(128, cfg.rope_max_seq_len, 31, 31, 45, 80)
    ref = O.rope_table(cfg, 31, 90, 160)  # complex128 [1, 1, N, 64]
    ang = torch.view_as_real(ref.reshape(-1, 64))
    assert torch.allclose(cs[..., 0].double(), ang[..., 0], atol=1e-6) and torch.allclose(cs[..., 1].double(), ang[..., 1], atol=1e-6)


def _build(cfg, params):
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    m = ChronoEditTransformer3DModel(
        num_attention_heads=cfg.num_attention_heads, attention_head_dim=cfg.attention_head_dim, in_channels=cfg.in_channels,
        out_channels=cfg.out_channels, text_dim=cfg.text_dim, freq_dim=cfg.freq_dim, ffn_dim=cfg.ffn_dim,
        num_layers=cfg.num_layers, image_dim=cfg.image_dim, added_kv_proj_dim=cfg.added_kv_proj_dim,
        rope_temporal_skip_len=cfg.rope_temporal_skip_len, device="cuda:0", dtype=torch.bfloat16)
    m.load_synthetic_({k: v.to("cuda:0") for k, v in params.items()})
    return m


@pytest.mark.parametrize("h,w,tag", [(132, 198, "configs[4] 1584x1056, N = 13068"), (90, 160, "configs[1] 1280x720, N = 7200")])
def test_full_width_block_bf16_and_fp8_modes_vs_fp32_oracle(h, w, tag):
    """One block of the 14B width (D = 5120, 40 heads, F = 13 824, 512 + 257 context rows) against the fp32 CPU oracle, in the bf16
    path and in the fp8 mode of bench.py --fp8 (e4m3 GEMMs with one scale per 5120- / 13824-long row + MXFP8 self-attention).
    Stated bounds: bf16 <= 1e-2; fp8 <= 10 x the bf16 path's error on the same input and <= 5e-2 absolute (measured on MI355X:
    bf16 4.76e-3, fp8 3.82e-2 = 8.0 x at both sizes; the bf16 HIP path sits on the reference's bf16-eager error:
    tests/test_dit_forward_gpu.py).  The 4 x the round-2 review asked for holds over depth (next test), not for a single block
    whose output is one residual update: an e4m3 x e4m3 product carries ~5 % rms noise, a bf16 one ~0.5 %."""
    cfg = O.DiTConfig(num_layers=1)
    p_bf = O.make_synthetic_params(cfg, seed=7, dtype=BF)
    lat, text, image = O.make_synthetic_inputs(cfg, 2, h, w, dtype=BF)
    model = _build(cfg, p_bf)
    ts = torch.tensor([800], device="cuda:0")
    args = (lat.cuda(), ts, text.cuda(), image.cuda())
    out_bf16 = model(*args, return_dict=False)[0].float().cpu()
    model.enable_fp8_gemms().enable_fp8_attention()           # the fp8 mode of bench.py --fp8: MX block scales on the GEMM operands (round 4)
    out_fp8 = model(*args, return_dict=False)[0].float().cpu()
    model.enable_fp8_gemms(policy="accurate")                 # round 6: the ungated cross-attention out-projection back on the bf16 GEMM
    out_acc = model(*args, return_dict=False)[0].float().cpu()
    singles = {}
    if h == 90:  # the per-Linear sensitivity table (tools/fp8_sensitivity.py runs this test with -s): ONE Linear in fp8 at a time,
        # attention in bf16; then only the MXFP8 self-attention
        for name in model.FP8_LINEARS:
            model.enable_fp8_gemms(linears=(name,)).enable_fp8_attention(False)
            singles[name] = model(*args, return_dict=False)[0].float().cpu()
        model.enable_fp8_gemms(False).enable_fp8_attention(True)
        singles["self-attention MXFP8"] = model(*args, return_dict=False)[0].float().cpu()
        model.enable_fp8_gemms(policy="fast").enable_fp8_attention(False)
        singles["all six, attention bf16"] = model(*args, return_dict=False)[0].float().cpu()
        model.enable_fp8_attention(True)
    model.enable_fp8_gemms(mx=False)                          # the round-1..3 contract: one scale per 5120- / 13824-long row
    out_row = model(*args, return_dict=False)[0].float().cpu()
    del model
    with torch.no_grad(), OD.on() as dev:  # the fp32 oracle, evaluated where oracle/device.py says (CE_ORACLE_DEVICE=cpu: the host cores)
        ref = O.dit_forward(OD.to(p_bf, dev, torch.float32), cfg, OD.to(lat, dev, torch.float32), torch.tensor([800]), OD.to(text, dev, torch.float32),
                            OD.to(image, dev, torch.float32)).cpu()
    e_bf16, e_fp8, e_row, e_acc = rel_l2(out_bf16, ref), rel_l2(out_fp8, ref), rel_l2(out_row, ref), rel_l2(out_acc, ref)
    print(f"full-width block, {tag}: bf16 path vs fp32 {e_bf16:.3e} | fp8 mode (MX block scales) vs fp32 {e_fp8:.3e} ({e_fp8 / e_bf16:.2f} x) | "
          f"per-row scales {e_row:.3e} ({e_row / e_bf16:.2f} x) | policy 'accurate' {e_acc:.3e} ({e_acc / e_bf16:.2f} x)")
    assert torch.isfinite(out_acc).all() and e_acc <= 5 * e_bf16, (e_acc, e_bf16)  # <= 5 x bf16 per block (measured 4.0 x)
    if singles:
        add = {}
        for name, o in singles.items():
            e = rel_l2(o, ref)
            add[name] = max(e * e - e_bf16 * e_bf16, 0.0) ** 0.5  # what this configuration ADDS, in quadrature, to the bf16 path's error
            print(f"   fp8 sensitivity, only {name:26s}: rel-L2 vs fp32 {e:.3e} = {e / e_bf16:5.2f} x bf16 | adds {add[name]:.3e}")
        six = sum(add[n] ** 2 for n in ("qkv", "o1", "q2", "o2", "f1", "f2")) ** 0.5
        print(f"   quadrature sum of the six single-Linear additions {six:.3e}; all six measured {add['all six, attention bf16']:.3e}")
        assert abs(six / add["all six, attention bf16"] - 1.0) < 0.1            # the contributions are independent: they add in quadrature
        assert add["o2"] ** 2 >= 0.6 * add["all six, attention bf16"] ** 2       # the ungated cross-attention out-projection dominates (measured 78 %)
        assert add["self-attention MXFP8"] <= 0.5 * e_bf16                      # the MXFP8 attention is not an error source at block level (1.0e-3)
    assert torch.isfinite(out_bf16).all() and torch.isfinite(out_fp8).all() and torch.isfinite(out_row).all()
    assert e_bf16 < 1e-2
    assert e_row <= 10 * e_bf16 and e_row < 5e-2, (e_row, e_bf16)
    assert e_fp8 <= e_row * 1.02 and e_fp8 <= 8 * e_bf16, (e_fp8, e_row, e_bf16)  # (bound tightened once measured: see DESIGN.md section 9)


def test_fp8_mode_error_growth_over_forty_blocks():
    """Depth of the real model at a narrow width, fp8 mode: the error against the fp32 oracle after 40 residual blocks stays
    within 4 x the bf16 path's on the same input and inside the per-forward tolerance of the fp8 mode (6e-2)."""
    cfg = O.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=40, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    p_bf = O.make_synthetic_params(cfg, seed=11, dtype=BF)
    lat, text, image = O.make_synthetic_inputs(cfg, 2, 16, 24, dtype=BF)
    model = _build(cfg, p_bf)
    ts = torch.tensor([700], device="cuda:0")
    args = (lat.cuda(), ts, text.cuda(), image.cuda())
    out_bf16 = model(*args, return_dict=False)[0].float().cpu()
    model.enable_fp8_gemms().enable_fp8_attention()
    out_fp8 = model(*args, return_dict=False)[0].float().cpu()
    p32 = {k: v.float() for k, v in p_bf.items()}
    with torch.no_grad():
        ref = O.dit_forward(p32, cfg, lat.float(), torch.tensor([700]), text.float(), image.float())
    e_bf16, e_fp8 = rel_l2(out_bf16, ref), rel_l2(out_fp8, ref)
    print(f"40 blocks: bf16 path vs fp32 {e_bf16:.3e} | fp8 mode vs fp32 {e_fp8:.3e} ({e_fp8 / e_bf16:.2f} x)")
    assert torch.isfinite(out_fp8).all()
    assert e_fp8 <= 4 * e_bf16 and e_fp8 < 6e-2, (e_fp8, e_bf16)
