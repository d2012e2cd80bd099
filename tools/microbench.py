"""Per-kernel timings on the GPU box (HIP events on the launch stream), with the vendor library
(torch.matmul -> hipBLASLt, torch SDPA) timed beside each kernel as a yardstick — not as a fallback.
Writes gpurun_out/microbench.json."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e-3


def main():
    res = {}
    only = sys.argv[1:] or ["gemm", "attn", "row"]
    g = torch.Generator().manual_seed(0)
    if "gemm" in only:
        for (M, N, K, epi) in [(7200, 15360, 5120, 0), (7200, 5120, 5120, 2), (7200, 13824, 5120, 1), (7200, 5120, 13824, 2),
                               (28800, 13824, 5120, 1), (512, 5120, 5120, 0)]:
            a = torch.randn(M, K, generator=g).to(BF).to(dev)
            w = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(dev)
            b = torch.zeros(N, device=dev)
            out = torch.empty(M, N, dtype=BF, device=dev)
            gate = torch.ones(N, device=dev)
            fl = 2.0 * M * N * K
            run = lambda: ops.gemm(a, w, b, out=out, epilogue=epi, gate=gate if epi == 2 else None, res=out if epi == 2 else None)
            ops.set_gemm_variant(0)
            t128 = timeit(run)
            ops.set_gemm_variant(2)
            tst = timeit(run)
            ops.set_gemm_variant(1)
            t = timeit(run)
            ops.set_gemm_variant(-1)
            t_ref = timeit(lambda: torch.matmul(a, w.t()))
            res[f"gemm_{M}x{N}x{K}_epi{epi}"] = {"ms": t * 1e3, "tflops": fl / t / 1e12, "tile128_tflops": fl / t128 / 1e12,
                                                 "hipblaslt_ms": t_ref * 1e3, "hipblaslt_tflops": fl / t_ref / 1e12}
            print(f"gemm {M}x{N}x{K} epi{epi}: tile256 {t*1e3:.3f} ms {fl/t/1e12:.1f} TF | staggered {fl/tst/1e12:.1f} TF | tile128 {fl/t128/1e12:.1f} TF | hipblaslt {t_ref*1e3:.3f} ms {fl/t_ref/1e12:.1f} TF", flush=True)
            del a, w, out
    if "split" in only:
        # split-K tail on/off for the shapes of one batched-CFG step (M = 2 x 7200) and the context GEMMs
        for (M, N, K, epi) in [(14400, 15360, 5120, 0), (14400, 5120, 5120, 2), (14400, 13824, 5120, 1), (14400, 5120, 13824, 2),
                               (7200, 15360, 5120, 0), (7200, 5120, 5120, 2), (7200, 13824, 5120, 1), (7200, 5120, 13824, 2),
                               (1538, 10240, 5120, 0), (28800, 5120, 5120, 2)]:
            a = torch.randn(M, K, generator=g).to(BF).to(dev)
            w = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(dev)
            b = torch.zeros(N, device=dev)
            out = torch.empty(M, N, dtype=BF, device=dev)
            gate = torch.ones(N, device=dev)
            fl = 2.0 * M * N * K
            run = lambda: ops.gemm(a, w, b, out=out, epilogue=epi, gate=gate if epi == 2 else None, res=out if epi == 2 else None)
            ops.set_gemm_variant(1)
            t0 = t1 = 1e9
            for _ in range(3):  # interleaved: clocks drift with temperature, so A/B/A/B and keep the best of each
                ops.set_gemm_split(False)
                t0 = min(t0, timeit(run, iters=20))
                ops.set_gemm_split(True)
                t1 = min(t1, timeit(run, iters=20))
            ops.set_gemm_variant(-1)
            tiles = ((M + 255) // 256) * ((N + 255) // 256)
            res[f"split_{M}x{N}x{K}_epi{epi}"] = {"whole_ms": t0 * 1e3, "split_ms": t1 * 1e3, "tiles": tiles}
            print(f"gemm {M}x{N}x{K} epi{epi}: tiles {tiles} (tail {tiles % 256}) whole {t0*1e3:.3f} ms {fl/t0/1e12:.1f} TF | split {t1*1e3:.3f} ms {fl/t1/1e12:.1f} TF", flush=True)
            del a, w, out
    if "fp8" in only:
        for (M, N, K, epi) in [(14400, 15360, 5120, 0), (14400, 5120, 5120, 2), (14400, 13824, 5120, 1), (14400, 5120, 13824, 2),
                               (7200, 15360, 5120, 0), (28800, 13824, 5120, 1)]:
            a = torch.randn(M, K, generator=g).to(BF).to(dev)
            w = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(dev)
            b = torch.zeros(N, device=dev)
            out = torch.empty(M, N, dtype=BF, device=dev)
            gate = torch.ones(N, device=dev)
            aq, sa = ops.quant_rows_fp8(a)
            wq, sw = ops.quant_rows_fp8(w)
            fl = 2.0 * M * N * K
            run8 = lambda: ops.gemm_fp8(aq, sa, wq, sw, b, out=out, epilogue=epi, gate=gate if epi == 2 else None, res=out if epi == 2 else None)
            run16 = lambda: ops.gemm(a, w, b, out=out, epilogue=epi, gate=gate if epi == 2 else None, res=out if epi == 2 else None)
            tq = timeit(lambda: ops.quant_rows_fp8(a, out=aq, scale=sa), iters=20)
            t8 = t16 = 1e9
            for _ in range(2):
                t16 = min(t16, timeit(run16, iters=10))
                t8 = min(t8, timeit(run8, iters=10))
            res[f"fp8_{M}x{N}x{K}_epi{epi}"] = {"fp8_ms": t8 * 1e3, "fp8_tflops": fl / t8 / 1e12, "bf16_ms": t16 * 1e3, "quant_a_ms": tq * 1e3}
            print(f"gemm {M}x{N}x{K} epi{epi}: fp8 {t8*1e3:.3f} ms {fl/t8/1e12:.1f} TF | bf16 {t16*1e3:.3f} ms {fl/t16/1e12:.1f} TF | "
                  f"activation quant pass {tq*1e3:.3f} ms ({3.0*M*K/tq/1e9:.0f} GB/s)", flush=True)
            del a, w, out, aq, wq
    if "attn" in only:
        for (Nq, Nkv, H) in [(7200, 7200, 40), (28800, 28800, 40), (7200, 512, 40)]:
            D = H * 128
            qkv = torch.randn(max(Nq, Nkv), 3 * D, generator=g).to(BF).to(dev)
            q, k, v = qkv[:Nq, :D], qkv[:Nkv, D:2 * D], qkv[:Nkv, 2 * D:]
            out = torch.empty(Nq, D, dtype=BF, device=dev)
            ops.set_attention_waves(4)
            t4 = timeit(lambda: ops.attention(q, k, v, H, out=out), iters=5)
            ops.set_attention_waves(8)
            t8 = timeit(lambda: ops.attention(q, k, v, H, out=out), iters=5)
            ops.set_attention_waves(64)
            t = timeit(lambda: ops.attention(q, k, v, H, out=out), iters=5)
            tvt = ttr = float("nan")
            if Nq == Nkv:
                vt = ops.v_transpose(v, H)
                tvt = timeit(lambda: ops.attention_vt(q, k, vt, H, out=out), iters=5)
                ttr = timeit(lambda: ops.v_transpose(v, H, out=vt), iters=5)
                print(f"attn {Nq}x{Nkv} H{H}: V^T by LDS-DMA {tvt*1e3:.3f} ms {4.0*Nq*Nkv*128*H/tvt/1e12:.1f} TF (+ transpose pass {ttr*1e3:.3f} ms) vs register-staged {t*1e3:.3f} ms", flush=True)
            ops.set_attention_waves(0)
            qh = q.reshape(Nq, H, 128).transpose(0, 1)[None].contiguous()
            kh = k.reshape(Nkv, H, 128).transpose(0, 1)[None].contiguous()
            vh = v.reshape(Nkv, H, 128).transpose(0, 1)[None].contiguous()
            try:
                t_ref = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qh, kh, vh), iters=5)
            except Exception as e:  # yardstick only
                print("sdpa yardstick failed:", e)
                t_ref = float("nan")
            if Nq == 7200:  # the two samples of a batched-CFG step in one launch
                qkv2 = torch.randn(2 * max(Nq, Nkv), 3 * D, generator=g).to(BF).to(dev)
                out2 = torch.empty(2 * Nq, D, dtype=BF, device=dev)
                if Nq == Nkv:
                    tb = timeit(lambda: ops.attention(qkv2[:, :D], qkv2[:, D:2 * D], qkv2[:, 2 * D:], H, out=out2, batch=2), iters=5)
                    vt2 = ops.v_transpose(qkv2[:, 2 * D:], H)
                    tb_vt = tb_tr = 1e9
                    for _ in range(3):  # interleaved, best of each
                        tb = min(tb, timeit(lambda: ops.attention(qkv2[:, :D], qkv2[:, D:2 * D], qkv2[:, 2 * D:], H, out=out2, batch=2), iters=5))
                        tb_vt = min(tb_vt, timeit(lambda: ops.attention_vt(qkv2[:, :D], qkv2[:, D:2 * D], vt2, H, out=out2, batch=2), iters=5))
                        tb_tr = min(tb_tr, timeit(lambda: ops.v_transpose(qkv2[:, 2 * D:], H, out=vt2), iters=5))
                    print(f"attn {Nq}x{Nkv} H{H} batch 2: V^T by LDS-DMA {tb_vt*1e3:.3f} ms {2*4.0*Nq*Nkv*128*H/tb_vt/1e12:.1f} TF (+ transpose pass {tb_tr*1e3:.3f} ms) vs register-staged {tb*1e3:.3f} ms {2*4.0*Nq*Nkv*128*H/tb/1e12:.1f} TF", flush=True)
                else:
                    q2 = qkv2[:2 * Nq, :D]
                    tb = timeit(lambda: ops.attention(q2, qkv2[:2 * Nkv, D:2 * D], qkv2[:2 * Nkv, 2 * D:], H, out=out2, batch=2), iters=5)
                print(f"attn {Nq}x{Nkv} H{H} batch 2 (auto kernel): {tb*1e3:.3f} ms {2*4.0*Nq*Nkv*128*H/tb/1e12:.1f} TF", flush=True)
                res[f"attn_{Nq}x{Nkv}x{H}_b2"] = {"ms": tb * 1e3, "tflops": 2 * 4.0 * Nq * Nkv * 128 * H / tb / 1e12}
                del qkv2, out2
            fl = 4.0 * Nq * Nkv * 128 * H
            res[f"attn_{Nq}x{Nkv}x{H}"] = {"ms": t * 1e3, "tflops": fl / t / 1e12, "sdpa_ms": t_ref * 1e3, "sdpa_tflops": fl / t_ref / 1e12}
            print(f"attn {Nq}x{Nkv} H{H}: plain 4-wave {fl/t4/1e12:.1f} TF | plain 8-wave {fl/t8/1e12:.1f} TF | software-pipelined {t*1e3:.3f} ms {fl/t/1e12:.1f} TF | torch sdpa {t_ref*1e3:.3f} ms {fl/t_ref/1e12:.1f} TF", flush=True)
            del qkv, qh, kh, vh
    if "attn8" in only:  # MXFP8 self-attention (three kernels) beside the bf16 kernel, per launch of a batched-CFG step
        for (N, H, B) in [(7200, 40, 2), (13068, 40, 2), (28800, 40, 1)]:
            D = H * 128
            qkv = torch.randn(B * N, 3 * D, generator=g).to(BF).to(dev)
            one = torch.ones(D, device=dev)
            out = torch.empty(B * N, D, dtype=BF, device=dev)
            q8, sq = ops.rmsnorm_rope_mxfp8(qkv[:, :D], one, None, 128, 1e-6, post_scale=ops.MXFP8_Q_SCALE)
            k8, sk = ops.rmsnorm_rope_mxfp8(qkv[:, D:2 * D], one, None, 128, 1e-6)
            v8t, sv = ops.v_mxfp8_transpose(qkv[:, 2 * D:], N, B, H)
            t_q = timeit(lambda: ops.rmsnorm_rope_mxfp8(qkv[:, :D], one, None, 128, 1e-6, out=q8, scale=sq), iters=10)
            t_v = timeit(lambda: ops.v_mxfp8_transpose(qkv[:, 2 * D:], N, B, H, out=v8t, scale=sv), iters=10)
            t_qb = timeit(lambda: ops.rmsnorm_rope_(qkv[:, :D], one, None, 128, 1e-6, x2=qkv[:, D:2 * D], w2=one), iters=10)
            t8 = t8p = t16 = 1e9
            for _ in range(3):
                t16 = min(t16, timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], H, out=out, batch=B), iters=3))
                ops.set_attention_mxfp8_variant(0)
                t8p = min(t8p, timeit(lambda: ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, out=out, batch=B), iters=3))
                ops.set_attention_mxfp8_variant(1)
                t8 = min(t8, timeit(lambda: ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, out=out, batch=B), iters=3))
            tpers = {}
            if True:
                ops.set_attention_mxfp8_variant(1)
                for nwg in (0, 256, 512, 768):
                    old = ops.set_attention_mxfp8_persistent(nwg)
                    tpers[nwg] = min(timeit(lambda: ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, out=out, batch=B), iters=3) for _ in range(3))
                    ops.set_attention_mxfp8_persistent(old)
                print(f"attn N={N} B={B} mxfp8 persistent workgroups: " + ", ".join(f"{k or 'per-item'}: {v*1e3:.3f} ms" for k, v in tpers.items()), flush=True)
            fl = 4.0 * N * N * 128 * H * B
            res[f"attn8_{N}x{H}_b{B}"] = {"mxfp8_ms": t8 * 1e3, "mxfp8_tflops": fl / t8 / 1e12, "bf16_ms": t16 * 1e3, "bf16_tflops": fl / t16 / 1e12,
                                          "qk_quant_ms_each": t_q * 1e3, "v_transpose_ms": t_v * 1e3, "bf16_norm_rope_qk_ms": t_qb * 1e3}
            print(f"attn N={N} H={H} B={B}: mxfp8 {t8*1e3:.3f} ms {fl/t8/1e12:.1f} TF (plain loop {t8p*1e3:.3f} ms {fl/t8p/1e12:.1f} TF) | bf16 {t16*1e3:.3f} ms {fl/t16/1e12:.1f} TF | producers: q or k norm+rope+quant "
                  f"{t_q*1e3:.3f} ms each, V^T quant {t_v*1e3:.3f} ms (bf16 path: q+k norm+rope {t_qb*1e3:.3f} ms)", flush=True)
            del qkv, out, q8, k8, v8t
    if "row" in only:
        M, D = 7200, 5120
        x = torch.randn(M, D, generator=g).to(BF).to(dev)
        y = torch.empty_like(x)
        a = torch.ones(D, device=dev)
        b = torch.zeros(D, device=dev)
        t = timeit(lambda: ops.ln_affine(x, a, b, 1e-6, out=y), iters=20)
        res["ln_affine_7200x5120"] = {"us": t * 1e6, "GBps": 2 * M * D * 2 / t / 1e9}
        cs = torch.rand(M, 64, 2, device=dev)
        t = timeit(lambda: ops.rmsnorm_rope_(x, a, cs, 128, 1e-6), iters=20)
        res["rmsnorm_rope_7200x5120"] = {"us": t * 1e6, "GBps": 2 * M * D * 2 / t / 1e9}
        print(res["ln_affine_7200x5120"], res["rmsnorm_rope_7200x5120"], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/microbench.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
