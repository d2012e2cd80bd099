"""The self-attention kernels on PEAKED score rows.  Both kernels run a speculative softmax - no row maximum on the common
path, an exact route per (wave, key tile) when a partial row sum outgrows the window (bf16: 2^10; MXFP8: the all-ones MFMA's overflow) - and every
timing in profiles/ so far used N(0, 0.02^2) weights, whose logits are ~N(0, 1).  Here the q norm weight scales the logits: sigma = 1 (the synthetic
statistics of bench.py) against sigma such that a row's maximum sits 8 ... 12 nats above its mean.  Per BASELINE configuration: rate of the bf16
(V^T / LDS-DMA) kernel and of the MXFP8 kernel (product library), and - from ONE launch each on the diagnostic library - the fraction of
(wave, tile) pairs that took the exact route.
    python tools/attn_peaked.py [sigma,sigma,...]      (default 1,2.4,3.0)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=5, rounds=3):
    best = 1e9
    for _ in range(rounds):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def main():
    sigmas = [float(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2.4,3.0").split(",")]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    H, hd = 40, 128
    D = H * hd
    for (N, B, what) in [(7200, 2, "configs[1] 720p, pair batched"), (7200, 1, "configs[2] 720p, B = 1"), (13068, 2, "configs[4] 1584x1056"),
                         (28800, 1, "configs[3] 8 latent frames (one sample)")]:
        M = N * B
        qkv0 = torch.randn(M, 3 * D, generator=g).to(BF).to(dev)
        one = torch.ones(D, device=dev)
        out = torch.empty(M, D, dtype=BF, device=dev)
        waves_tiles = B * H * ((N + 255) // 256) * 8 * ((N + 63) // 64)  # (wave, key tile) pairs of the 8-wave kernels
        for sg in sigmas:
            qkv = qkv0.clone()
            wq = torch.full((D,), sg, device=dev)
            # bf16 operands: RMSNorm across heads (unit rms per channel), q scaled by sigma -> logits q.k / sqrt(128) ~ N(0, sigma^2)
            ops.rmsnorm_rope_(qkv[:, :D], wq, None, hd, 1e-6, x2=qkv[:, D:2 * D], w2=one)
            vt = ops.v_transpose(qkv[:, 2 * D:], H)
            # measured peak of a sample of rows (head 0, 64 rows, sample 0): row max - row mean in nats
            qs, ks = qkv[:64, :hd].float(), qkv[:N, D:D + hd].float()
            lg = qs @ ks.t() / hd ** 0.5
            peak = float((lg.max(dim=1).values - lg.mean(dim=1)).mean())
            t16 = timeit(lambda: ops.attention_vt(qkv[:, :D], qkv[:, D:2 * D], vt, H, out=out, batch=B))
            # MXFP8 operands from the same pre-norm tensors
            q8, sq = ops.rmsnorm_rope_mxfp8(qkv0[:, :D], wq, None, hd, 1e-6, post_scale=ops.MXFP8_Q_SCALE)
            k8, sk = ops.rmsnorm_rope_mxfp8(qkv0[:, D:2 * D], one, None, hd, 1e-6)
            v8t, sv = ops.v_mxfp8_transpose(qkv0[:, 2 * D:], N, B, H)
            t8 = timeit(lambda: ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, out=out, batch=B))
            # exact-route counts: one launch each through the diagnostic library
            ops.force_diagnostics(True)
            ops.attention_exact_route_hits(reset=True)
            ops.attention_vt(qkv[:, :D], qkv[:, D:2 * D], vt, H, out=out, batch=B)
            h16, _ = ops.attention_exact_route_hits(reset=True)
            ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, out=out, batch=B)
            _, h8 = ops.attention_exact_route_hits(reset=True)
            ops.force_diagnostics(False)
            fl = 4.0 * N * N * hd * H * B
            print(f"{what:42s} N={N:6d} B={B} sigma {sg:3.1f} (row max - mean = {peak:4.1f} nats): bf16 {t16:.3f} ms {fl / t16 / 1e9:6.0f} TF, exact route "
                  f"{h16 / waves_tiles * 100:6.3f} % of (wave, tile) | MXFP8 {t8:.3f} ms {fl / t8 / 1e9:6.0f} TF, exact route {h8 / waves_tiles * 100:6.3f} %", flush=True)
            del qkv, vt, q8, k8, v8t
        del qkv0, out


if __name__ == "__main__":
    main()
