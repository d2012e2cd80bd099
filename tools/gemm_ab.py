"""A/B timing of builds of the GEMM kernels in ONE process (see tools/attn_ab.py for the why):
    python tools/gemm_ab.py <base.so> [<variant.so> ...]
Each library is `hipcc -shared` of ce_gemm.hip + ce_gemm256.hip (plus -D switches); all are called through ce_gemm_bf16 on
the same tensors, interleaved, best of each; the outputs are compared with the first library's."""
import ctypes
import sys

import torch

BF = torch.bfloat16


def bind(path):
    lib = ctypes.CDLL(path)
    P, I = ctypes.c_void_p, ctypes.c_int
    f = lib.ce_gemm_bf16
    f.restype = I
    f.argtypes = [P, P, P, P, I, P, P] + [I] * 8 + [P]
    lib.ce_set_gemm_workspace.argtypes = [P, ctypes.c_size_t]
    return lib, f


def main():
    libs = [bind(p) for p in sys.argv[1:]]
    names = [p.split("/")[-1].replace("lib", "").replace(".so", "") for p in sys.argv[1:]]
    dev = torch.device("cuda:0")
    ws = torch.empty(96 * 1024 * 1024, dtype=torch.uint8, device=dev)  # the engine's scratch size (ops.GEMM_WS_BYTES)
    for lib, _ in libs:
        lib.ce_set_gemm_workspace(ws.data_ptr(), ws.numel())
    g = torch.Generator().manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K, epi) in [(14400, 10240, 5120, 0), (14400, 13824, 5120, 1), (14400, 5120, 13824, 2), (14400, 5120, 5120, 2), (14400, 5120, 5120, 0), (5120, 14400, 5120, 6),
                           (7200, 10240, 5120, 0), (7200, 13824, 5120, 1), (7200, 5120, 13824, 2), (7200, 5120, 5120, 2), (7200, 5120, 5120, 0), (5120, 7200, 5120, 6),
                           (26136, 13824, 5120, 1), (28800, 5120, 5120, 2)]:
        a = torch.randn(M, K, generator=g).to(BF).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(dev)
        b = torch.randn(max(M, N), generator=g).to(dev)
        gate = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(BF).to(dev)
        outs = [torch.empty(M, N, dtype=BF, device=dev) for _ in libs]

        def run(f, o):
            rc = f(a.data_ptr(), w.data_ptr(), o.data_ptr(), b.data_ptr(), epi, gate.data_ptr() if epi == 2 else None,
                   res.data_ptr() if epi == 2 else None, M, N, K, K, K, N, N, 0, st)
            assert rc == 0, rc

        def timeit(f, o, iters=10):
            run(f, o)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run(f, o)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters

        best = [1e9] * len(libs)
        for _ in range(4):
            for i, (_, f) in enumerate(libs):
                best[i] = min(best[i], timeit(f, outs[i]))
        fl = 2.0 * M * N * K
        line = f"gemm {M}x{N}x{K} epi{epi}:"
        for i, n in enumerate(names):
            d = (outs[0].float() - outs[i].float()).norm().item() / outs[0].float().norm().item()
            line += f" | {n} {best[i]:.3f} ms {fl/best[i]/1e9:.0f} TF ({(best[0]/best[i]-1)*100:+.1f} %, rel {d:.1e})"
        print(line, flush=True)
        del a, w, res, outs


if __name__ == "__main__":
    main()
