"""A/B of operand layouts for the 256-tile GEMM in ONE process: row-major (K contiguous rows, pitch K) vs K-slab-major
([K/64][rows][64]: each 16 KiB LDS-DMA half-tile is one contiguous block) for W only / A only / both.  Results are compared
with the row-major run (must be bit-identical: same arithmetic, different addresses)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

BF = torch.bfloat16


def slab(x):  # [R, K] -> [K/64, R, 64] contiguous
    R, K = x.shape
    return x.view(R, K // 64, 64).permute(1, 0, 2).contiguous()


def main():
    dev = torch.device("cuda:0")
    lib = ops.lib()
    ops.ensure_gemm_workspace(dev)
    g = torch.Generator().manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K, epi) in [(14400, 15360, 5120, 0), (14400, 13824, 5120, 1), (14400, 5120, 13824, 2), (14400, 5120, 5120, 2)]:
        a = torch.randn(M, K, generator=g).to(BF).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        gate = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(BF).to(dev)
        a_s, w_s = slab(a), slab(w)
        outs = {}

        def run(name, out):
            A, lda, ask, ass = (a_s, 64, 64, M * 64) if "A" in name else (a, K, 0, 0)
            W, ldw, wsk, wss = (w_s, 64, 64, N * 64) if "W" in name else (w, K, 0, 0)
            rc = lib.ce_gemm_seg_bf16(A.data_ptr(), W.data_ptr(), out.data_ptr(), b.data_ptr(), epi, gate.data_ptr() if epi == 2 else None,
                                      res.data_ptr() if epi == 2 else None, M, N, K, lda, ldw, N, N, 0, ask, ass, wsk, wss, st)
            assert rc == 0, (name, rc)

        def timeit(name, out, iters=10):
            run(name, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run(name, out)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters

        names = ["rowmajor", "slabW", "slabA", "slabAW"]
        best = {n: 1e9 for n in names}
        for n in names:
            outs[n] = torch.empty(M, N, dtype=BF, device=dev)
        for _ in range(4):
            for n in names:
                best[n] = min(best[n], timeit(n, outs[n]))
        fl = 2.0 * M * N * K
        line = f"gemm {M}x{N}x{K} epi{epi}:"
        for n in names:
            same = torch.equal(outs[n], outs["rowmajor"])
            line += f" | {n} {best[n]:.3f} ms {fl/best[n]/1e9:.0f} TF ({(best['rowmajor']/best[n]-1)*100:+.1f} %{'' if same else ' MISMATCH'})"
        print(line, flush=True)
        del a, w, res, outs, a_s, w_s


if __name__ == "__main__":
    main()
