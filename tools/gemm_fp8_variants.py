"""A/B of the two main loops of ce_gemm_fp8 in one process (ce_set_gemm_fp8_variant: 0 = 8 waves / 4 phases, 1 = one wave per SIMD) on
the step's shapes, outputs compared bit for bit.   python tools/gemm_fp8_variants.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    shapes = [(14400, 15360, 5120, ops.EPI_BIAS, "qkv"), (14400, 5120, 5120, ops.EPI_GATE_RES, "out-proj"),
              (14400, 13824, 5120, ops.EPI_BIAS_GELU, "ffn-up"), (14400, 5120, 13824, ops.EPI_GATE_RES, "ffn-down"),
              (26136, 13824, 5120, ops.EPI_BIAS_GELU, "ffn-up 1584x1056")]
    tot = {0: 0.0, 1: 0.0}
    for (M, N, K, epi, tag) in shapes:
        a = torch.randn(M, K, generator=g).to(BF).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(dev)
        aq, sa = ops.quant_rows_fp8(a)
        wq, sw = ops.quant_rows_fp8(w)
        b = torch.randn(N, generator=g).to(dev)
        gate = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(BF).to(dev)
        outs = {v: torch.empty(M, N, dtype=BF, device=dev) for v in (0, 1)}
        kw = dict(epilogue=epi)
        if epi == ops.EPI_GATE_RES:
            kw.update(gate=gate, res=res)

        def timeit(v, iters=10):
            ops.set_gemm_fp8_variant(v)
            ops.gemm_fp8(aq, sa, wq, sw, b, out=outs[v], **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.gemm_fp8(aq, sa, wq, sw, b, out=outs[v], **kw)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters

        best = {0: 1e9, 1: 1e9}
        for _ in range(rounds):
            for v in (0, 1):
                best[v] = min(best[v], timeit(v))
        ops.set_gemm_fp8_variant(1)
        fl = 2.0 * M * N * K
        same = bool(torch.equal(outs[0], outs[1]))
        print(f"gemm_fp8 {tag:18s} {M}x{N}x{K} epi{epi}: w8 {best[0]:.3f} ms {fl / best[0] / 1e9:.0f} TF | w4 {best[1]:.3f} ms {fl / best[1] / 1e9:.0f} TF "
              f"({(best[0] / best[1] - 1) * 100:+.1f} %), bit-identical {same}", flush=True)
        tot[0] += best[0]
        tot[1] += best[1]
    print("sum (ms):", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
