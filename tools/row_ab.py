"""A/B timing of builds of the row kernels (ce_rowops.hip) in ONE process:  python tools/row_ab.py <base.so> [<variant.so> ...]
Shapes of one batched-CFG step: LN-modulate of [14400, 5120] and RMSNorm + RoPE of the q / k thirds of a [14400, 15360] buffer."""
import ctypes
import sys

import torch

BF = torch.bfloat16


def bind(path):
    lib = ctypes.CDLL(path)
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.ce_ln_affine_bf16.argtypes = [P, P, P, P, I, I, I, I, F, I, I, P]
    lib.ce_rmsnorm_rope_bf16.argtypes = [P, P, P, P, P, I, I, I, I, F, I, P]
    return lib


def main():
    libs = [bind(p) for p in sys.argv[1:]]
    names = [p.split("/")[-1].replace("lib", "").replace(".so", "") for p in sys.argv[1:]]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    M, D = 14400, 5120
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(M, D, generator=g).to(BF).to(dev)
    ab = torch.randn(2, 2, D, generator=g).to(dev)  # per-sample (a, b)
    qkv0 = torch.randn(M, 3 * D, generator=g).to(BF).to(dev)
    w = (1 + 0.1 * torch.randn(2, D, generator=g)).to(dev)
    cs = torch.randn(7200, 64, 2, generator=g).to(dev)
    outs = [torch.empty(M, D, dtype=BF, device=dev) for _ in libs]
    qkvs = [qkv0.clone() for _ in libs]

    def timeit(fn, iters=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    def ln(i):
        a, b = ab[:, 0].contiguous(), ab[:, 1].contiguous()
        return lambda: libs[i].ce_ln_affine_bf16(x.data_ptr(), outs[i].data_ptr(), a.data_ptr(), b.data_ptr(), M, D, D, D, 1e-6, 7200, D, st)

    def rr(i):
        q = qkvs[i]
        return lambda: libs[i].ce_rmsnorm_rope_bf16(q.data_ptr(), w[0].data_ptr(), q.data_ptr() + 2 * D, w[1].data_ptr(), cs.data_ptr(), M, D,
                                                      3 * D, 128, 1e-6, 7200, st)

    for label, mk, bytes_ in (("ln_affine 14400x5120", ln, 2 * M * D * 2), ("rmsnorm_rope 14400x5120 x2", rr, 2 * 2 * M * D * 2)):
        fns = [mk(i) for i in range(len(libs))]
        best = [1e9] * len(libs)
        for _ in range(4):
            for i, f in enumerate(fns):
                best[i] = min(best[i], timeit(f))
        line = label + ":"
        for i, n in enumerate(names):
            line += f" | {n} {best[i]:.1f} us {bytes_/best[i]/1e6:.2f} TB/s ({(best[0]/best[i]-1)*100:+.1f} %)"
        print(line, flush=True)
    # results: LN outputs must agree bit for bit; the in-place RMSNorm+RoPE is applied once to fresh copies
    for i in range(len(libs)):
        qkvs[i].copy_(qkv0)
        rr(i)()
    torch.cuda.synchronize()
    for i, n in enumerate(names[1:], 1):
        d = (outs[0].float() - outs[i].float()).abs()
        print(f"{n}: ln equal {torch.equal(outs[0], outs[i])} (differing elements {float((d > 0).float().mean()):.2e}, max |d| {float(d.max()):.3e}, "
              f"max |d| / |value| {float((d / outs[0].float().abs().clamp_min(1e-3)).max()):.3e}), rmsnorm_rope equal {torch.equal(qkvs[0], qkvs[i])}")


if __name__ == "__main__":
    main()
