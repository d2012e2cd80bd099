"""Per-rank compute of ONE transformer block at the shapes a rank of a W-GPU Ulysses group runs (configs[3], N = 28 800 tokens),
measured on one GPU: the row-sharded GEMMs and row passes at rows = B * ceil64(N / W), the head-sharded self-attention at H / W heads
over all tokens (blocked layout for B = 2), cross-attention on the local rows.  No exchange is emulated - tools/scaling_model.py
adds the wire.  Usage: python tools/rank_emulation.py  ->  one line per (W, B) with the per-block milliseconds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, D, F, H, Tt, Ti = 28800, 5120, 13824, 40, 512, 257


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rnd(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(BF).to(dev)


w_qkv, w_o, w_q2, w_f1, w_f2 = rnd(3 * D, D, scale=.02), rnd(D, D, scale=.02), rnd(D, D, scale=.02), rnd(F, D, scale=.02), rnd(D, F, scale=.02)
b3, b1, bf = torch.zeros(3 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(F, device=dev)
gate = torch.ones(D, device=dev)
ln_a, ln_b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
for W in (1, 4, 8):
    for B in ((2,) if W == 1 else (1, 2)):
        n = (N + W - 1) // W
        if B == 2 and W > 1:
            n = (n + 63) // 64 * 64
        rows, hl, Dl = B * n, H // W, D // W
        x, h, qkv, att, ffn = rnd(rows, D), rnd(rows, D), rnd(rows, 3 * D), rnd(rows, D), rnd(rows, F)
        cs = torch.rand(n, 64, 2, device=dev)
        t = {}
        t["ln x3"] = 3 * timeit(lambda: ops.ln_affine(x, ln_a, ln_b, 1e-6, out=h))
        t["kv+q proj"] = timeit(lambda: ops.gemm(h, w_qkv[D:], b3[D:], out=qkv[:, D:])) + timeit(lambda: ops.gemm(h, w_qkv[:D], b3[:D], out=qkv[:, :D]))
        t["norm+rope"] = timeit(lambda: ops.rmsnorm_rope_(qkv[:, :D], ln_a, cs, 128, 1e-6, x2=qkv[:, D:2 * D], w2=ln_a))
        # head-sharded self-attention over ALL tokens (W * n rows per sample incl. padding), hl heads
        T = W * n
        q_s, k_s = rnd(B * T, hl * 128), rnd(B * T, hl * 128)
        out_s = torch.empty(B * T, hl * 128, dtype=BF, device=dev)
        if W == 1:
            vt = ops.v_transpose(rnd(B * T, hl * 128), hl)
            t["self-attn"] = timeit(lambda: ops.attention_vt(q_s, k_s, vt, hl, out=out_s, batch=B), iters=3)
        elif B == 1:  # one sample: the receive buffer is plain [global token]; keys = the N valid tokens
            vsrc = rnd(N, hl * 128)
            vt = ops.v_transpose(vsrc, hl)
            t["self-attn"] = timeit(lambda: ops.attention_vt(q_s, k_s[:N], vt, hl, out=out_s), iters=3) + timeit(lambda: ops.v_transpose(vsrc, hl, out=vt))
        else:
            vsrc = rnd(B * T, hl * 128)
            vt = ops.v_transpose_blocked(vsrc, hl, B, n, N)
            t["self-attn"] = timeit(lambda: ops.attention_vt_blocked(q_s, k_s, vt, hl, B, n, N, out=out_s), iters=3) + \
                timeit(lambda: ops.v_transpose_blocked(vsrc, hl, B, n, N, out=vt))
        t["o1, q2, o2"] = 2 * timeit(lambda: ops.gemm(att, w_o, b1, out=x, epilogue=ops.EPI_GATE_RES, gate=gate, res=x)) + \
            timeit(lambda: ops.gemm(h, w_q2, b1, out=att))
        kv_t, kv_i = rnd(B * Tt, 2 * D), rnd(B * Ti, 2 * D)
        t["cross-attn"] = timeit(lambda: ops.attention(att, kv_t[:, :D], kv_t[:, D:], H, out=h, k2=kv_i[:, :D], v2=kv_i[:, D:], batch=B))
        t["ffn"] = timeit(lambda: ops.gemm(h, w_f1, bf, out=ffn, epilogue=ops.EPI_BIAS_GELU)) + \
            timeit(lambda: ops.gemm(ffn, w_f2, b1, out=x, epilogue=ops.EPI_GATE_RES, gate=gate, res=x))
        tot = sum(t.values())
        passes = 2 // B
        print(f"W={W} B={B}: rows {rows}, heads {hl}: block {tot:.3f} ms x {passes} pass(es) = {passes * tot:.3f} ms per guidance step and layer | "
              + ", ".join(f"{k} {v:.3f}" for k, v in t.items()), flush=True)
        del x, h, qkv, att, ffn, q_s, k_s, out_s, vt
