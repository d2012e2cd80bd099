"""Launch ONE kernel shape a few times (for rocprofv3 --pmc passes).  usage:
   one_kernel.py gemm M N K epi variant [iters]  |  one_kernel.py attn Nq Nkv H [iters]  |  one_kernel.py attn8 N H B [iters]
   |  one_kernel.py attnvt N H B [iters]   (the V^T / LDS-DMA form of the bf16 self-attention, B samples per launch)
   |  one_kernel.py gemm8 M N K epi [iters]   (MX fp8 GEMM, ce_gemm_mxfp8: operands quantised once outside the counted launches; epi 7 = the
      FFN-up form with the GELU + MX quantiser in the epilogue, ce_gemm_mxfp8_gelu_quant)
   |  one_kernel.py conv KT Cin Cout T H W n_tile [iters]   (ce_conv3d_gemm_bf16; n_tile 1 = the slab kernel of the 96-channel layers)
   |  one_kernel.py attn1 N C [iters]   (the VAE mid-block attention, ce_attention_1head_bf16)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
kind = sys.argv[1]
if kind == "gemm":
    M, N, K, epi, var = map(int, sys.argv[2:7])
    iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(dev)
    b = torch.zeros(N, device=dev)
    out = torch.zeros((N, M) if epi == 7 else (M, N), dtype=BF, device=dev)  # (epi 7 = CE_EPI_BIAS_T: the transpose is stored)
    gate = torch.ones(N, device=dev)
    ops.set_gemm_variant(var)
    for _ in range(iters):
        ops.gemm(a, w, b, out=out, epilogue=epi, gate=gate if epi == 2 else None, res=out if epi == 2 else None)
elif kind == "gemm8":
    M, N, K, epi = map(int, sys.argv[2:6])
    iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    a = torch.randn(M, K, generator=g).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(dev)
    aq, sa = ops.quant_rows_mxfp8(a)
    wq, sw = ops.quant_rows_mxfp8(w, w_order=True)
    b = torch.zeros(N, device=dev)
    gate = torch.ones(N, device=dev)
    if epi == 7:
        oq = torch.empty(M, N, dtype=torch.uint8, device=dev)
        so = torch.empty(ops.mx_scale_bytes(M, N), dtype=torch.uint8, device=dev)
        for _ in range(iters):
            ops.gemm_mxfp8_gelu_quant(aq, sa, wq, sw, b, oq, so)
    else:
        out = torch.zeros(M, N, dtype=BF, device=dev)
        for _ in range(iters):
            ops.gemm_mxfp8(aq, sa, wq, sw, b, out=out, epilogue=epi, gate=gate if epi == 2 else None, res=out if epi == 2 else None)
elif kind == "conv":
    from chronoedit_amd.vae import Frames, _ConvPack
    KT, Cin, Cout, T, H, W, n_tile = map(int, sys.argv[2:9])
    iters = int(sys.argv[9]) if len(sys.argv) > 9 else 5
    n_in = T + KT - 1
    f = Frames(T, H, W, Cin, dev, front=KT - 1)
    f.stack[:n_in, 1:-1, 1:-1] = torch.randn(n_in, H, W, Cin, generator=g).to(BF).to(dev)
    pk = _ConvPack((torch.randn(Cout, Cin, KT, 3, 3, generator=g) / (9 * KT * Cin) ** 0.5).to(BF).to(dev), torch.randn(Cout, generator=g).to(dev))
    out = Frames(T, H, W, Cout, dev)
    for _ in range(iters):
        ops.conv3d_gemm(f.stack, pk.gemm_weight(), pk.b, out.data, None, T_out=T, H=H, W=W, Cin=Cin, Cout=Cout, KT=KT, n_tile=n_tile)
elif kind == "attn1":
    N, C = map(int, sys.argv[2:4])
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    qkv = torch.randn(N, 3 * C, generator=g).to(BF).to(dev)
    vt = torch.zeros((C, (N + 63) // 64 * 64), dtype=BF, device=dev)
    vt[:, :N] = qkv[:, 2 * C:].t()
    for _ in range(iters):
        ops.attention_1head(qkv[:, :C], qkv[:, C:2 * C], vt, C ** -0.5)
elif kind == "attnvt":
    N, H, B = map(int, sys.argv[2:5])
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    D = H * 128
    qkv = torch.randn(B * N, 3 * D, generator=g).to(BF).to(dev)
    vt = ops.v_transpose(qkv[:, 2 * D:], H)
    out = torch.empty(B * N, D, dtype=BF, device=dev)
    ops.set_attention_waves(int(os.environ.get("CE_ATTN_WAVES", "0")))  # 0: the 8-wave body; 128 / 129: one wave per SIMD
    for _ in range(iters):
        ops.attention_vt(qkv[:, :D], qkv[:, D:2 * D], vt, H, out=out, batch=B)
elif kind == "attn8":  # MXFP8 self-attention, B samples per launch (producers run once, outside the counted launches' names)
    N, H, B = map(int, sys.argv[2:5])
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    D = H * 128
    qkv = torch.randn(B * N, 3 * D, generator=g).to(BF).to(dev)
    one = torch.ones(D, device=dev)
    q8, sq = ops.rmsnorm_rope_mxfp8(qkv[:, :D], one, None, 128, 1e-6, post_scale=ops.MXFP8_Q_SCALE)
    k8, sk = ops.rmsnorm_rope_mxfp8(qkv[:, D:2 * D], one, None, 128, 1e-6)
    v8t, sv = ops.v_mxfp8_transpose(qkv[:, 2 * D:], N, B, H)
    out = torch.empty(B * N, D, dtype=BF, device=dev)
    for _ in range(iters):
        ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, out=out, batch=B)
else:
    Nq, Nkv, H = map(int, sys.argv[2:5])
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    D = H * 128
    ops.set_attention_waves(int(os.environ.get("CE_ATTN_WAVES", "0")))
    qkv = torch.randn(max(Nq, Nkv), 3 * D, generator=g).to(BF).to(dev)
    out = torch.empty(Nq, D, dtype=BF, device=dev)
    for _ in range(iters):
        ops.attention(qkv[:Nq, :D], qkv[:Nkv, D:2 * D], qkv[:Nkv, 2 * D:], H, out=out)
torch.cuda.synchronize()
