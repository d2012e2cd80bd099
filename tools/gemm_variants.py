"""A/B of the main loops of the 256-tile bf16 GEMM in ONE process (ce_set_gemm_variant: 1 = 8 waves / 8 phases, 2 = staggered,
3 / 4 = one wave per SIMD with an A ring of 3 / 2 stages) on the shapes of one batched-CFG denoising step (M = 2 x 7200) and of the
N = 28 800 mode; random operands (zero-filled ones clock ~20 % higher on this chip), interleaved rounds, best and median of each.
    python tools/gemm_variants.py [variants, e.g. 1,3,4] [rounds]
CE_GEMM_AB_M=7200: the B = 1 shapes of the distilled configuration (BASELINE configs[2]); CE_GEMM_WS_MB=n: split-K scratch of n MiB (default 64)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

BF = torch.bfloat16
NAMES = {0: "tile128", 1: "w8", 2: "w8stag", 3: "w4-3st", 4: "w4", 5: "w4-1bar", 6: "384x256"}


def main():
    variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,4,3,5").split(",")]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    if os.environ.get("CE_GEMM_WS_MB"):
        ops.GEMM_WS_BYTES = int(os.environ["CE_GEMM_WS_MB"]) << 20
    MM = int(os.environ.get("CE_GEMM_AB_M", "14400"))
    shapes = [(14400, 10240, 5120, ops.EPI_BIAS, "q|k"), (5120, 14400, 5120, ops.EPI_BIAS_ROW, "V^T"),
              (14400, 5120, 5120, ops.EPI_GATE_RES, "out-proj"), (14400, 13824, 5120, ops.EPI_BIAS_GELU, "ffn-up"),
              (14400, 5120, 13824, ops.EPI_GATE_RES, "ffn-down"), (28800, 13824, 5120, ops.EPI_BIAS_GELU, "ffn-up 28800")]
    if MM != 14400:
        shapes = [(MM if m == 14400 else m, MM if n == 14400 else n, k, e, t) for (m, n, k, e, t) in shapes[:5]] + [(MM, 5120, 5120, ops.EPI_BIAS, "cross q")]
    tot = {v: 0.0 for v in variants}
    for (M, N, K, epi, tag) in shapes:
        a = torch.randn(M, K, generator=g).to(BF).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(dev)
        b = torch.randn(M if epi == ops.EPI_BIAS_ROW else N, generator=g).to(dev)
        gate = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(BF).to(dev)
        outs = {v: torch.empty(M, N, dtype=BF, device=dev) for v in variants}
        kw = dict(epilogue=epi)
        if epi == ops.EPI_GATE_RES:
            kw.update(gate=gate, res=res)

        def timeit(v, iters=10):
            ops.set_gemm_variant(v)
            ops.gemm(a, w, b, out=outs[v], **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.gemm(a, w, b, out=outs[v], **kw)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters

        times = {v: [] for v in variants}
        for _ in range(rounds):
            for v in variants:
                times[v].append(timeit(v))
        ops.set_gemm_variant(-1)
        fl = 2.0 * M * N * K
        line = f"gemm {tag:13s} {M}x{N}x{K} epi{epi}:"
        base = variants[0]
        for v in variants:
            best, med = min(times[v]), statistics.median(times[v])
            d = (outs[base].float() - outs[v].float()).norm().item() / outs[base].float().norm().item()
            line += f" | {NAMES.get(v, v)} best {best:.3f} ms {fl/best/1e9:.0f} TF, median {fl/med/1e9:.0f} TF ({(min(times[base])/best-1)*100:+.1f} %, rel {d:.1e})"
            if M in (14400, MM) or M == 5120:
                tot[v] += best * (2 if tag == "out-proj" else 1)  # two 5120x5120 projections per block
        print(line, flush=True)
        del a, w, res, outs
    print("sum over one block's large GEMMs (ms):", {NAMES.get(v, v): round(t, 3) for v, t in tot.items()}, flush=True)


if __name__ == "__main__":
    main()
