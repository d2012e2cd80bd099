"""GPU parity tests for every entry point of libchronoedit_hip.so, called through the C ABI
(chronoedit_amd.ops -> ctypes).  The reference for each op is plain fp32 PyTorch evaluating the
same formula as the oracle (oracle/dit_oracle.py), with the reference's bf16 rounding points.
Tolerances are stated per test: bf16 outputs are compared at <= 2 bf16 ulp-ish relative error
(2^-7) element-wise on normalised data, or rel-L2 for accumulated results."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_library_loaded_is_in_tree():
    from chronoedit_amd import hiplib, ops
    lib = ops.lib()
    assert hiplib.LIB_PATH.endswith("chronoedit_amd/lib/libchronoedit_hip.so")
    for name in hiplib.header_symbols():
        assert hasattr(lib, name)


@pytest.mark.parametrize("M,D", [(37, 256), (200, 5120), (7200, 5120)])
def test_ln_affine(M, D):
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(M, D, generator=g) * 3 + 0.5).to(BF).to(dev)
    a = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    b = (0.1 * torch.randn(D, generator=g)).to(dev)
    y = ops.ln_affine(x, a, b, 1e-6)
    ref = (torch.nn.functional.layer_norm(x.float(), (D,), None, None, 1e-6) * a + b).to(BF)
    err = (y.float() - ref.float()).abs().max().item()
    assert err <= 2 ** -6 * ref.float().abs().max().item(), err
    assert rel_l2(y, ref) < 3e-3


@pytest.mark.parametrize("M,H,rope", [(50, 2, True), (300, 40, True), (257, 40, False)])
def test_rmsnorm_rope(M, H, rope):
    from chronoedit_amd import ops
    dev = _dev()
    D = H * 128
    g = torch.Generator().manual_seed(1)
    buf = torch.randn(M, 3 * D, generator=g).to(BF).to(dev)  # strided view like the fused qkv buffer
    before = buf.clone()
    x = buf[:, D:2 * D]
    x0 = x.clone()
    w = (1 + 0.05 * torch.randn(D, generator=g)).to(BF).to(dev)
    cs = None
    if rope:
        ang = torch.rand(M, 64, generator=g, dtype=torch.float64) * 6.28
        cs = torch.stack([ang.cos(), ang.sin()], -1).float().to(dev)
    ops.rmsnorm_rope_(x, w.float(), cs, 128, 1e-6)
    # reference: diffusers RMSNorm rounding points, RoPE in fp64 (transformer_chronoedit.py:73-76)
    xf = x0.float()
    y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(BF) * w
    if rope:
        yc = torch.view_as_complex(y.double().view(M, H, 64, 2))
        f = torch.view_as_complex(cs.double()).view(M, 1, 64)
        y = torch.view_as_real(yc * f).reshape(M, D).to(BF)
    assert torch.equal(buf[:, :D], before[:, :D]) and torch.equal(buf[:, 2 * D:], before[:, 2 * D:])  # neighbours untouched
    assert rel_l2(x, y) < 4e-3, rel_l2(x, y)
    frac_exact = (x == y).float().mean().item()
    assert frac_exact > 0.97, frac_exact  # fp32-vs-fp64 rope differs only at bf16 rounding ties


GEMM_SHAPES = [
    (128, 128, 64), (100, 264, 192), (7200, 5120, 5120), (769, 10240, 5120), (512, 5120, 4096), (333, 64, 5120),
]


@pytest.fixture(params=[0, 1, 2, 3, 4, 5, 6, 7], ids=["tile128", "tile256w8", "tile256w8stag", "tile256w4_3stage", "tile256w4", "tile256w4_1barrier", "tile384x256", "tile288x256"])
def gemm_variant(request):
    from chronoedit_amd import ops
    old = ops.set_gemm_variant(request.param)
    yield request.param
    ops.set_gemm_variant(old)


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES + [(256, 256, 128), (7200, 13824, 5120), (1000, 520, 13824)])
def test_gemm_bias(M, N, K, gemm_variant):
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
    # asymmetric, transpose-detecting content
    w[: min(N, 7), :] *= 3
    bias = torch.randn(N, generator=g).to(dev)
    out = ops.gemm(a, w, bias)
    ref = (a.float() @ w.float().t() + bias)
    assert rel_l2(out, ref) < 4e-3, rel_l2(out, ref)
    assert (out.float() - ref).abs().max().item() <= 2 ** -6 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("M,N,K", [(128, 136, 64), (5120, 1000, 1024), (640, 14400, 512), (5120, 7200, 5120)])
def test_gemm_row_bias_is_the_transpose_of_the_column_bias_product(M, N, K, gemm_variant):
    """CE_EPI_BIAS_ROW (bias along the rows of C: the product with its operand roles swapped, V^T = W_v.X^T) == the transpose of
    the ordinary product, bit for bit where both run the same kernel (one rounding point: bf16(acc + bias)); also into a wider,
    strided output (the V^T buffer with its padding columns, which must stay untouched)."""
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    wv = (torch.randn(M, K, generator=g) * 0.05).to(BF).to(dev)   # "weights": rows of C
    x = torch.randn(N, K, generator=g).to(BF).to(dev)             # "tokens": columns of C
    bias = torch.randn(M, generator=g).to(dev)
    buf = torch.full((M, N + 72), 7.0, dtype=BF, device=dev)
    ops.gemm(wv, x, bias, out=buf[:, :N], epilogue=ops.EPI_BIAS_ROW)
    ref = wv.float() @ x.float().t() + bias[:, None]
    assert rel_l2(buf[:, :N], ref) < 4e-3
    assert (buf[:, N:] == 7.0).all()
    plain = ops.gemm(x, wv, bias)  # [N, M] = X.W^T + bias[col]
    assert rel_l2(buf[:, :N], plain.t()) < 1e-4  # same products and rounding point; the fp32 summation order may differ (split-K tail tiles)


@pytest.mark.parametrize("variant", [-1, 4, 6, 7], ids=["auto", "tile256w4", "tile384x256", "tile288x256"])
@pytest.mark.parametrize("M,N,K", [(14400, 5120, 5120), (7200, 5120, 5120), (128, 136, 64), (1000, 520, 1024), (5120, 1024, 512)])
def test_gemm_transposed_store_is_the_row_bias_product_with_swapped_operands(M, N, K, variant):
    """CE_EPI_BIAS_T (the transpose of the product is stored: V^T = (X.W_v^T)^T with M = tokens) == CE_EPI_BIAS_ROW with the operand roles
    swapped (V^T = W_v.X^T), bit for bit where neither cuts a split-K tail - the same products in the same k order; into a wider, strided
    output whose padding columns must stay untouched; the two step shapes run the 384- / 288-row kernel's transposed store, the small ones
    (and every forced variant but 6 / 7) the swapped product itself."""
    from chronoedit_amd import ops
    dev = _dev()
    old_variant = ops.set_gemm_variant(variant)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(M, K, generator=g).to(BF).to(dev)              # tokens
    wv = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)    # weights
    wv[: min(N, 5)] *= 3
    bias = torch.randn(N, generator=g).to(dev)
    buf = torch.full((N, M + 72), 7.0, dtype=BF, device=dev)
    ops.gemm(x, wv, bias, out=buf[:, :M], epilogue=ops.EPI_BIAS_T)
    ref = (x.float() @ wv.float().t() + bias).t()
    assert rel_l2(buf[:, :M], ref) < 4e-3
    assert (buf[:, M:] == 7.0).all()
    old = torch.full((N, M + 72), 7.0, dtype=BF, device=dev)
    ops.gemm(wv, x, bias, out=old[:, :M], epilogue=ops.EPI_BIAS_ROW)
    ops.set_gemm_variant(old_variant)
    assert rel_l2(buf[:, :M], old[:, :M]) < 1e-4  # (not always bit-equal: the swapped product may have cut a split-K tail - another fp32 summation order)


def test_gemm_epilogues(gemm_variant):
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    M, N, K = 300, 384, 256
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.08).to(BF).to(dev)
    bias = torch.randn(N, generator=g).to(dev) * 0.2
    lin = (a.float() @ w.float().t() + bias).to(BF)
    out = ops.gemm(a, w, bias, epilogue=ops.EPI_BIAS_GELU)
    ref = torch.nn.functional.gelu(lin.float(), approximate="tanh").to(BF)
    assert rel_l2(out, ref) < 5e-3
    out = ops.gemm(a, w, bias, epilogue=ops.EPI_BIAS_GELU_ERF)
    ref = torch.nn.functional.gelu(lin.float()).to(BF)
    assert rel_l2(out, ref) < 5e-3
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    gate = torch.randn(N, generator=g).to(dev)
    x = res.clone()
    ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_GATE_RES, gate=gate, res=x)  # in place, as the block uses it
    ref = (res.float() + lin.float() * gate).to(BF)
    assert rel_l2(x, ref) < 5e-3
    x = res.clone()
    ops.gemm(a, w, bias, out=x, epilogue=ops.EPI_GATE_RES, gate=None, res=x)
    ref = (res.float() + lin.float()).to(BF)
    assert rel_l2(x, ref) < 5e-3


@pytest.mark.parametrize("variant", [1, 4, 6, 7], ids=["tile256w8", "tile256w4", "tile384x256", "tile288x256"])
@pytest.mark.parametrize("M,N,K,epi", [(4352, 4096, 1024, "bias"), (7200, 5120, 5120, "gate"), (1538, 10240, 5120, "gelu"),
                                       (600, 512, 1024, "bias")])
def test_gemm_split_k_tail(M, N, K, epi, variant):
    """Tail tiles of the 256-tile kernel are cut along K (fp32 slabs + reduce launch); same result as running whole."""
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.03).to(BF).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(BF).to(dev)
    gate = torch.randn(N, generator=g).to(dev)
    kw = {"bias": dict(), "gelu": dict(epilogue=ops.EPI_BIAS_GELU),
          "gate": dict(epilogue=ops.EPI_GATE_RES, gate=gate, res=res)}[epi]
    old_v = ops.set_gemm_variant(variant)
    try:
        old_s = ops.set_gemm_split(False)
        whole = ops.gemm(a, w, bias, **kw)
        ops.set_gemm_split(True)
        split = ops.gemm(a, w, bias, **kw)
        split2 = ops.gemm(a, w, bias, **kw)
        ops.set_gemm_split(old_s)
    finally:
        ops.set_gemm_variant(old_v)
    assert torch.equal(split, split2)                         # fixed reduction order: deterministic
    lin = a.float() @ w.float().t() + bias
    ref = {"bias": lin, "gelu": torch.nn.functional.gelu(lin.to(BF).float(), approximate="tanh"),
           "gate": res.float() + lin.to(BF).float() * gate}[epi]
    assert rel_l2(split, ref) < 4e-3 and rel_l2(whole, ref) < 4e-3
    # only fp32 summation order differs between the two paths: at most one bf16 ulp on a few elements
    d = (split.float() - whole.float()).abs()
    assert d.max().item() <= 2 ** -7 * ref.abs().max().item()
    assert (d > 0).float().mean().item() < 0.05


def test_gemm_rejects_bad_shapes():
    from chronoedit_amd import ops
    dev = _dev()
    a = torch.zeros(8, 100, dtype=BF, device=dev)
    w = torch.zeros(16, 100, dtype=BF, device=dev)
    with pytest.raises(ops.HipKernelError):
        ops.gemm(a, w, None)
    with pytest.raises(ops.HipKernelError):
        ops.gemm(a.cpu(), w.cpu(), None)


def _sdpa_ref(q, k, v, H):
    Nq, D = q.shape
    qh = q.float().view(Nq, H, 128).transpose(0, 1)
    kh = k.float().view(-1, H, 128).transpose(0, 1)
    vh = v.float().view(-1, H, 128).transpose(0, 1)
    o = torch.nn.functional.scaled_dot_product_attention(qh[None], kh[None], vh[None])[0]
    return o.transpose(0, 1).reshape(Nq, D)


@pytest.fixture(params=[0, 64, 8, 4], ids=["auto", "swpipe", "wg8", "wg4"])
def attn_waves(request):
    from chronoedit_amd import ops
    old = ops.set_attention_waves(request.param)
    yield request.param
    ops.set_attention_waves(old)


@pytest.mark.parametrize("two_seg", [False, True])
def test_attention_batched_equals_per_sample(two_seg, attn_waves):
    """batch samples stacked along rows, one launch == one launch per sample (bit-exact)."""
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(14)
    B, Nq, L1, L2, H = 3, 333, 200 if two_seg else 333, 77, 8
    D = H * 128
    q = torch.randn(B * Nq, D, generator=g).to(BF).to(dev)
    kv1 = torch.randn(B * L1, 2 * D, generator=g).to(BF).to(dev)
    kv2 = torch.randn(B * L2, 2 * D, generator=g).to(BF).to(dev)
    kw = dict(k2=kv2[:, :D], v2=kv2[:, D:]) if two_seg else {}
    out = ops.attention(q, kv1[:, :D], kv1[:, D:], H, batch=B, **kw)
    for b in range(B):
        kwb = dict(k2=kv2[b * L2:(b + 1) * L2, :D], v2=kv2[b * L2:(b + 1) * L2, D:]) if two_seg else {}
        ref = ops.attention(q[b * Nq:(b + 1) * Nq], kv1[b * L1:(b + 1) * L1, :D], kv1[b * L1:(b + 1) * L1, D:], H, **kwb)
        assert torch.equal(out[b * Nq:(b + 1) * Nq], ref), b
    with pytest.raises(ValueError):
        ops.attention(q[:-1], kv1[:, :D], kv1[:, D:], H, batch=B)


@pytest.mark.parametrize("Nq,Nkv,H,B", [(256, 256, 8, 1), (512, 100, 8, 2), (290, 64, 5, 3), (31, 700, 16, 2), (1056, 1056, 5, 2),
                                         (769, 769, 24, 1)])
def test_attention_work_order_shapes(Nq, Nkv, H, B):
    """Default kernel over the corner cases of its workgroup order: no remainder block, only a remainder block, head counts that
    are / are not multiples of 8 (XCD-aware vs plain mapping), several stacked samples."""
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    D = H * 128
    q = torch.randn(B * Nq, D, generator=g).to(BF).to(dev)
    kv = torch.randn(B * Nkv, 2 * D, generator=g).to(BF).to(dev)
    out = ops.attention(q, kv[:, :D], kv[:, D:], H, batch=B)
    for b in range(B):
        ref = _sdpa_ref(q[b * Nq:(b + 1) * Nq], kv[b * Nkv:(b + 1) * Nkv, :D], kv[b * Nkv:(b + 1) * Nkv, D:], H)
        assert rel_l2(out[b * Nq:(b + 1) * Nq], ref) < 1e-2, (b, rel_l2(out[b * Nq:(b + 1) * Nq], ref))


@pytest.mark.parametrize("Nq,Nkv,H", [(64, 64, 2), (300, 257, 2), (1000, 1000, 8), (7200, 7200, 8), (33, 512, 3)])
def test_attention_single_segment(Nq, Nkv, H, attn_waves):
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    D = H * 128
    qkv = torch.randn(max(Nq, Nkv), 3 * D, generator=g).to(BF).to(dev)
    q, k, v = qkv[:Nq, :D], qkv[:Nkv, D:2 * D], qkv[:Nkv, 2 * D:]
    # make v asymmetric across dv so a transposed/permuted PV shows up
    v.mul_(torch.linspace(0.5, 1.5, D, device=dev).to(BF))
    out = ops.attention(q, k, v, H)
    ref = _sdpa_ref(q, k, v, H)
    assert rel_l2(out, ref) < 1e-2, rel_l2(out, ref)
    assert (out.float() - ref).abs().max().item() < 3e-2


@pytest.fixture(params=[0, 128, 129], ids=["sp-2-waves-per-simd", "w4-1-wave-per-simd", "w4-persistent"])
def vt_body(request):
    """Loop body of the V^T form (ce_set_attention_waves): the 8-wave software-pipelined kernel or the one-wave-per-SIMD kernel (round 4)."""
    from chronoedit_amd import ops
    old = ops.set_attention_waves(request.param)
    yield request.param
    ops.set_attention_waves(old)


# (round 6: the one-wave-per-SIMD body is a closed alternative of round 4, selectable in the diagnostic library only - four shapes x two slopes)
@pytest.mark.parametrize("Nq,Nkv,H,B", [(300, 257, 2, 1), (7200, 7200, 8, 2), (290, 64, 5, 3), (31, 704, 16, 2)])
@pytest.mark.parametrize("slope", [0.0, 10.0])
def test_attention_vt_one_wave_per_simd_body_is_bit_identical_to_the_eight_wave_body(Nq, Nkv, H, B, slope):
    """attn_fwd_w4_kernel (4 waves x 64 query rows, Q and O^T in the accumulator file, every matrix instruction an asm statement) computes
    the SAME products in the SAME order with the same rounding points as attn_fwd_sp_kernel<false, true>: outputs are equal bit for bit -
    on random scores (speculative softmax all the way), with a spike growing 0.5 / 10 octaves per key tile (offset moved on the exact
    route, in every tile at slope 10), with query blocks whose second sub-block or later waves have no rows, and with key tails."""
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(31)
    D = H * 128
    q = torch.randn(B * Nq, D, generator=g).to(BF).to(dev)
    k = torch.randn(B * Nkv, D, generator=g).to(BF).to(dev)
    v = torch.randn(B * Nkv, D, generator=g).to(BF).to(dev)
    if slope:
        for b in range(B):
            for t in range((Nkv + 63) // 64):
                k[b * Nkv + min(t * 64 + 5, Nkv - 1)] = q[b * Nq + min(7, Nq - 1)] * (0.5 + slope * t)
    vt = ops.v_transpose(v, H)
    outs = {}
    for body in (0, 128, 129):
        old = ops.set_attention_waves(body)
        try:
            outs[body] = ops.attention_vt(q, k, vt, H, batch=B).clone()
        finally:
            ops.set_attention_waves(old)
    assert torch.isfinite(outs[128].float()).all()
    assert torch.equal(outs[128], outs[0]), (outs[128].float() - outs[0].float()).abs().max()
    assert torch.equal(outs[129], outs[0]), (outs[129].float() - outs[0].float()).abs().max()
    for b in range(B):
        ref = _sdpa_ref(q[b * Nq:(b + 1) * Nq], k[b * Nkv:(b + 1) * Nkv], v[b * Nkv:(b + 1) * Nkv], H)
        # (spiked rows put nearly all of their weight on one key per tile: 1.08e-2 measured at 7200 keys and slope 10 - for BOTH bodies)
        assert rel_l2(outs[128][b * Nq:(b + 1) * Nq], ref) < (1e-2 if slope == 0 else 1.5e-2)


# (round 6: a measured-and-closed experiment that lives in the DIAGNOSTIC library only - three shapes keep it honest: ragged, the step's key
# count, and unaligned sample offsets where the default body must take over)
@pytest.mark.parametrize("Nq,Nkv,H,B,slope", [(300, 257, 2, 1, 0.0), (7200, 7200, 8, 2, 10.0), (1090, 1090, 8, 2, 0.0)])
def test_attention_vt_16x16x32_body(Nq, Nkv, H, B, slope):
    """attn_fwd_x16_kernel (csrc/ce_attn16.hip, round 5; ce_set_attention_waves(16)): the V^T self-attention on v_mfma_f32_16x16x32_bf16 - K rows
    permuted inside the tile so that a lane's P operand meets one 16-byte read of natural-order V^T, online softmax with a lazily moved offset -
    against torch SDPA (<= 1e-2; spiked scores, which move the offset in every tile, <= 1.5e-2) and against the default 32x32x16 body (<= 3e-3:
    same products, other summation orders); query blocks with empty waves, key tails, many work items per persistent workgroup."""
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(37)
    D = H * 128
    q = torch.randn(B * Nq, D, generator=g).to(BF).to(dev)
    k = torch.randn(B * Nkv, D, generator=g).to(BF).to(dev)
    v = torch.randn(B * Nkv, D, generator=g).to(BF).to(dev)
    v.mul_(torch.linspace(0.5, 1.5, D, device=dev).to(BF)).add_((torch.arange(B * Nkv, device=dev) % 7).to(BF)[:, None] * 0.25)  # a permuted P.V shows up
    if slope:
        for b in range(B):
            for t in range((Nkv + 63) // 64):
                k[b * Nkv + min(t * 64 + 5, Nkv - 1)] = q[b * Nq + min(7, Nq - 1)] * (0.5 + slope * t)
    vt = ops.v_transpose(v, H)
    base = ops.attention_vt(q, k, vt, H, batch=B).clone()
    old = ops.set_attention_waves(16)
    try:
        out = ops.attention_vt(q, k, vt, H, batch=B).clone()
    finally:
        ops.set_attention_waves(old)
    assert torch.isfinite(out.float()).all()
    if slope == 0:
        assert rel_l2(out, base) < 3e-3, rel_l2(out, base)
    for b in range(B):
        ref = _sdpa_ref(q[b * Nq:(b + 1) * Nq], k[b * Nkv:(b + 1) * Nkv], v[b * Nkv:(b + 1) * Nkv], H)
        e = rel_l2(out[b * Nq:(b + 1) * Nq], ref)
        assert e < (1e-2 if slope == 0 else 1.5e-2), (b, e, rel_l2(base[b * Nq:(b + 1) * Nq], ref))


@pytest.mark.parametrize("Nq,Nkv,H,B", [(64, 64, 2, 1), (300, 257, 2, 1), (1000, 1000, 8, 1), (512, 104, 8, 2), (290, 64, 5, 3), (31, 704, 16, 2),
                                         (7200, 7200, 8, 2), (1056, 1056, 5, 2), (1090, 1090, 8, 2), (330, 3270, 8, 3),   # these two: sample offsets only 4-byte aligned
                                         (2000, 200, 40, 3), (3000, 64, 24, 2)])  # many work items per persistent workgroup
def test_attention_vt_matches_sdpa_and_the_register_staged_kernel(Nq, Nkv, H, B, vt_body):
    """ce_attention_vt_bf16 (V handed over transposed by ce_v_transpose_bf16; K rows permuted inside the tile so that a lane's
    keys are contiguous in V^T): vs torch SDPA <= 1e-2, vs the register-staged kernel <= 3e-3 (same products, the row sums and
    the P.V contraction run over the keys of a tile in a different order); the padding columns of V^T are zero."""
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(23)
    D = H * 128
    q = torch.randn(B * Nq, D, generator=g).to(BF).to(dev)
    kv = torch.randn(B * Nkv, 2 * D, generator=g).to(BF).to(dev)
    kv[:, D:].mul_(torch.linspace(0.5, 1.5, D, device=dev).to(BF))  # asymmetric across dv and ...
    kv[:, D:].add_((torch.arange(B * Nkv, device=dev) % 7).to(BF)[:, None] * 0.25)  # ... across keys: a permuted P.V shows up
    vt = ops.v_transpose(kv[:, D:], H)
    assert vt.shape == (D, ops.vt_columns(B * Nkv))
    assert torch.equal(vt[:, :B * Nkv], kv[:, D:].t()) and not vt[:, B * Nkv:].any()
    out = ops.attention_vt(q, kv[:, :D], vt, H, batch=B)
    base = ops.attention(q, kv[:, :D], kv[:, D:], H, batch=B)
    assert rel_l2(out, base) < 3e-3, rel_l2(out, base)
    for b in range(B):
        ref = _sdpa_ref(q[b * Nq:(b + 1) * Nq], kv[b * Nkv:(b + 1) * Nkv, :D], kv[b * Nkv:(b + 1) * Nkv, D:], H)
        assert rel_l2(out[b * Nq:(b + 1) * Nq], ref) < 1e-2, (b, rel_l2(out[b * Nq:(b + 1) * Nq], ref))


@pytest.mark.parametrize("N,n,W,H,B", [(300, 128, 3, 2, 2), (128, 64, 2, 8, 1), (1000, 256, 4, 8, 2), (7100, 3584, 2, 8, 2), (500, 64, 8, 5, 3)])
def test_attention_vt_blocked_layout_equals_plain(N, n, W, H, B, vt_body):
    """ce_attention_vt_blocked_bf16 + ce_v_transpose_blocked_bf16: q / k / v / out rows in the all-to-all receive layout
    [source rank][sample][local token] (n tokens per rank, W ranks, the last block padded: N valid tokens) == the plain-layout
    kernel on the un-blocked tensors, bit for bit (same kernel, only row addresses differ)."""
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(29)
    D, T = H * 128, W * n
    assert (W - 1) * n < N <= T
    plain = torch.randn(B, T, 3 * D, generator=g).to(BF).to(dev)     # [sample][token][q | k | v]
    plain[:, N:, D:] = 0.5                                             # padded keys: finite junk the mask must hide
    blocked = plain.view(B, W, n, 3 * D).permute(1, 0, 2, 3).contiguous().view(W * B * n, 3 * D)  # [src][b][i]
    vt = ops.v_transpose_blocked(blocked[:, 2 * D:], H, B, n, N)
    cols = vt.shape[1] // B
    for b in range(B):
        assert torch.equal(vt[:, b * cols:b * cols + N], plain[b, :N, 2 * D:].t()) and not vt[:, b * cols + N:(b + 1) * cols].any()
    out = ops.attention_vt_blocked(blocked[:, :D], blocked[:, D:2 * D], vt, H, B, n, N)
    got = out.view(W, B, n, D).permute(1, 0, 2, 3).reshape(B, T, D)
    for b in range(B):
        vt_b = ops.v_transpose(plain[b, :N, 2 * D:], H)
        want = ops.attention_vt(plain[b, :, :D], plain[b, :N, D:2 * D], vt_b, H)
        assert torch.equal(got[b], want), (b, (got[b].float() - want.float()).abs().max())
        ref = _sdpa_ref(plain[b, :, :D], plain[b, :N, D:2 * D], plain[b, :N, 2 * D:], H)
        assert rel_l2(got[b], ref) < 1e-2


@pytest.mark.parametrize("slope", [0.5, 10.0])
def test_attention_vt_spiked_scores_take_the_exact_route(slope, vt_body):
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    H, N = 2, 520
    q = torch.randn(N, H * 128, generator=g).to(BF).to(dev)
    k = torch.randn(N, H * 128, generator=g).to(BF).to(dev)
    v = torch.randn(N, H * 128, generator=g).to(BF).to(dev)
    for t in range(N // 64):
        k[t * 64 + 5] = q[7] * (0.5 + slope * t)
    out = ops.attention_vt(q, k, ops.v_transpose(v, H), H)
    ref = _sdpa_ref(q, k, v, H)
    assert rel_l2(out, ref) < 1e-2 and torch.isfinite(out.float()).all()


@pytest.mark.parametrize("slope", [0.5, 10.0])
def test_attention_spiked_scores_force_rescale(attn_waves, slope):
    """One key per tile has a much larger score than everything before it: exercises the online-softmax
    rescale path on every tile (cdna guide rule 26).  slope 0.5: +8 octaves per tile (the default kernel's speculative
    softmax takes its exact route every second tile, and runs with P up to 2^8 in between); slope 10: +160 octaves per
    tile, exp2 overflows to inf before the row-sum check sends the tile through the exact route."""
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    H, N = 2, 512
    q = torch.randn(N, H * 128, generator=g).to(BF).to(dev)
    k = torch.randn(N, H * 128, generator=g).to(BF).to(dev)
    v = torch.randn(N, H * 128, generator=g).to(BF).to(dev)
    for t in range(N // 64):
        k[t * 64 + 5] = q[7] * (0.5 + slope * t)
    out = ops.attention(q, k, v, H)
    ref = _sdpa_ref(q, k, v, H)
    assert rel_l2(out, ref) < 1e-2
    assert torch.isfinite(out.float()).all()


def test_attention_two_segments(attn_waves):
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(6)
    H, Nq = 4, 700
    D = H * 128
    q = torch.randn(Nq, D, generator=g).to(BF).to(dev)
    kv_t = torch.randn(512, 2 * D, generator=g).to(BF).to(dev)
    kv_i = torch.randn(257, 2 * D, generator=g).to(BF).to(dev)
    out = ops.attention(q, kv_t[:, :D], kv_t[:, D:], H, k2=kv_i[:, :D], v2=kv_i[:, D:])
    ref = (_sdpa_ref(q, kv_t[:, :D], kv_t[:, D:], H).to(BF).float() + _sdpa_ref(q, kv_i[:, :D], kv_i[:, D:], H).to(BF).float())
    assert rel_l2(out, ref) < 1e-2, rel_l2(out, ref)


@pytest.mark.parametrize("Nq,L1,L2,H,B", [(700, 512, 257, 4, 1), (300, 512, 257, 2, 2), (513, 100, 65, 3, 2), (7200, 512, 257, 8, 2)])
def test_attention_two_segments_vt_form(Nq, L1, L2, H, B):
    """ce_attention_2seg_vt_bf16 (cross-attention with both V operands transposed, K and V^T tiles by LDS-DMA) vs fp32 SDPA per segment
    and vs the register-staged form the engine runs; batches with padded per-sample column strides (257 keys -> 320 columns), key counts
    that end inside a tile, a remainder query block."""
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(Nq + L2)
    D = H * 128
    q = torch.randn(B * Nq, D, generator=g).to(BF).to(dev)
    kv1 = torch.randn(B * L1, 2 * D, generator=g).to(BF).to(dev)
    kv2 = torch.randn(B * L2, 2 * D, generator=g).to(BF).to(dev)
    kv2[:, D:].add_((torch.arange(B * L2, device=dev) % 5).to(BF)[:, None] * 0.5)  # key-dependent v: a permuted P.V shows up

    def vt_of(v, ln):
        cols = (ln + 63) // 64 * 64
        vt = torch.zeros((D, B * cols), dtype=BF, device=dev)
        for b in range(B):
            vt[:, b * cols : b * cols + ln] = v[b * ln : (b + 1) * ln].t()
        return vt

    out = ops.attention_2seg_vt(q, kv1[:, :D], vt_of(kv1[:, D:], L1), L1, kv2[:, :D], vt_of(kv2[:, D:], L2), L2, H, batch=B)
    old = ops.attention(q, kv1[:, :D], kv1[:, D:], H, k2=kv2[:, :D], v2=kv2[:, D:], batch=B)
    assert rel_l2(out, old) < 4e-3, rel_l2(out, old)  # same products, another key order inside a tile
    for b in range(B):
        qb = q[b * Nq : (b + 1) * Nq]
        ref = (_sdpa_ref(qb, kv1[b * L1 : (b + 1) * L1, :D], kv1[b * L1 : (b + 1) * L1, D:], H).to(BF).float()
               + _sdpa_ref(qb, kv2[b * L2 : (b + 1) * L2, :D], kv2[b * L2 : (b + 1) * L2, D:], H).to(BF).float())
        assert rel_l2(out[b * Nq : (b + 1) * Nq], ref) < 1e-2, (b, rel_l2(out[b * Nq : (b + 1) * Nq], ref))


def test_timestep_chain_and_modulation():
    from chronoedit_amd import ops
    from oracle import dit_oracle as O
    dev = _dev()
    t = torch.tensor([637], device=dev)
    s = ops.timestep_sinusoid(t, 256)
    ref = O.timestep_sinusoid(torch.tensor([637]), 256)[0]
    assert (s.cpu() - ref).abs().max().item() < 2e-4  # fp32 sin/cos of arguments up to 637 rad
    g = torch.Generator().manual_seed(7)
    W = torch.randn(96, 256, generator=g).to(dev)
    b = torch.randn(96, generator=g).to(dev)
    y = ops.gemv(W, s, b, flags=2)
    yr = torch.nn.functional.silu(W @ s + b)
    assert rel_l2(y, yr) < 1e-5
    Wb = W.to(BF)
    y = ops.gemv(Wb, s, b, flags=1 | 4)
    xin = torch.nn.functional.silu(s).to(BF).float()
    yr = (Wb.float() @ xin + b).to(BF).float()
    assert rel_l2(y, yr) < 2e-3
    table = torch.randn(3, 6, 64, generator=g).to(dev)
    v = torch.randn(6, 64, generator=g).to(dev)
    m = ops.modulation(table, v, 0b010010)
    r = table + v
    r[:, 1] += 1
    r[:, 4] += 1
    assert torch.allclose(m, r, atol=1e-6)


def test_patchify_unpatchify_roundtrip_against_conv():
    from chronoedit_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(8)
    C, T, H, W, D = 36, 2, 12, 20, 128
    x = torch.randn(C, T, H, W, generator=g).to(BF).to(dev)
    wt = (torch.randn(D, C, 1, 2, 2, generator=g) * 0.1).to(BF).to(dev)
    cols = ops.patchify(x, 192)
    wp = torch.zeros(D, 192, dtype=BF, device=dev)
    wp[:, :144] = wt.reshape(D, -1)
    y = ops.gemm(cols, wp, None)
    ref = torch.nn.functional.conv3d(x[None].float(), wt.float(), stride=(1, 2, 2)).flatten(2).transpose(1, 2)[0]
    assert rel_l2(y, ref) < 4e-3
    # unpatchify == reshape/permute of transformer_chronoedit.py:463-467
    Cout = 16
    z = torch.randn(T * (H // 2) * (W // 2), 64, generator=g).to(BF).to(dev)
    out = ops.unpatchify(z, Cout, T, H, W)
    r = z.reshape(1, T, H // 2, W // 2, 1, 2, 2, Cout).permute(0, 7, 1, 4, 2, 5, 3, 6).flatten(6, 7).flatten(4, 5).flatten(2, 3)[0]
    assert torch.equal(out, r)
