"""The automatic macro-tile choice of ce_gemm_bf16 (ce_gemm_bf16_tile_rows: 384 x 256, 288 x 256 or 256 x 256) against a measurement of ALL THREE tiles, for the
large GEMMs of a block at every row count the engine runs them on: M = 7 200 (BASELINE configs[2], B = 1), 14 400 (configs[1], pair batched),
13 068 / 26 136 (configs[4]), 28 800 / 57 600 (configs[3] on one GPU), 3 648 / 7 296 (a rank of the 8-GPU Ulysses split, B = 1 / 2).
Interleaved rounds, best of each; "loss" = what the model's pick costs against the measured winner.
    python tools/gemm_tile_choice.py [M,M,...] [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    Ms = [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else "7200,14400,13068,26136,28800,57600,3648,7296").split(",")]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    lib = ops.lib()
    tot_auto = tot_best = 0.0
    for MM in Ms:
        shapes = [(MM, 10240, 5120, ops.EPI_BIAS, "q|k"), (5120, (MM + 63) // 64 * 64, 5120, ops.EPI_BIAS_ROW, "V^T"), (MM, 5120, 5120, ops.EPI_GATE_RES, "out-proj"),
                  (MM, 13824, 5120, ops.EPI_BIAS_GELU, "ffn-up"), (MM, 5120, 13824, ops.EPI_GATE_RES, "ffn-down")]
        for (M, N, K, epi, tag) in shapes:
            a = torch.randn(M, K, generator=g).to(BF).to(dev)
            w = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(dev)
            b = torch.randn(M if epi == ops.EPI_BIAS_ROW else N, generator=g).to(dev)
            kw = dict(epilogue=epi)
            if epi == ops.EPI_GATE_RES:
                kw.update(gate=torch.randn(N, generator=g).to(dev), res=torch.randn(M, N, generator=g).to(BF).to(dev))
            out = torch.empty(M, N, dtype=BF, device=dev)

            def timeit(v, iters=8):
                ops.set_gemm_variant(v)
                ops.gemm(a, w, b, out=out, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    ops.gemm(a, w, b, out=out, **kw)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / iters

            t = {4: 1e9, 6: 1e9, 7: 1e9}
            outs = {}
            for _ in range(rounds):
                for v in (4, 6, 7):
                    t[v] = min(t[v], timeit(v))
                    outs[v] = out.clone()
            same = bool(torch.equal(outs[7], outs[6]))  # the 288-row form of the 384-row kernel: the same k order per accumulator
            ops.set_gemm_variant(-1)
            pick = lib.ce_gemm_bf16_tile_rows(M, N, K, 256, ops.GEMM_WS_BYTES)
            t_pick, t_best = t[{384: 6, 288: 7, 256: 4}[pick]], min(t.values())
            tot_auto += t_pick
            tot_best += t_best
            fl = 2.0 * M * N * K
            best_v = min(t, key=t.get)
            print(f"{tag:9s} {M:6d}x{N:5d}x{K:5d}: 256-row {t[4]:.3f} ms {fl/t[4]/1e9:5.0f} TF | 288-row {t[7]:.3f} ms {fl/t[7]/1e9:5.0f} TF (== 384-row result: {same}) | "
                  f"384-row {t[6]:.3f} ms {fl/t[6]/1e9:5.0f} TF | measured {({4: 256, 6: 384, 7: 288})[best_v]} | model picks {pick}" +
                  ("" if t_pick == t_best else f"  <-- loses {(t_pick/t_best-1)*100:.1f} %"), flush=True)
            del a, w, out, kw
    print(f"sum over all shapes: automatic choice {tot_auto:.3f} ms, per-shape best {tot_best:.3f} ms ({(tot_auto/tot_best-1)*100:.2f} % lost to wrong picks)")


if __name__ == "__main__":
    main()
