"""A/B timing of builds of the V^T / LDS-DMA attention kernel in ONE process (see tools/attn_ab.py):
    python tools/attn_vt_ab.py <base.so> [<variant.so> ...]
Each library is `hipcc -shared [-D...]` of ce_attn.hip; all run ce_attention_vt_bf16 on the same tensors, interleaved, best of each."""
import ctypes
import sys

import torch

BF = torch.bfloat16


def bind(path):
    lib = ctypes.CDLL(path)
    P, I = ctypes.c_void_p, ctypes.c_int
    f = lib.ce_attention_vt_bf16
    f.restype = I
    f.argtypes = [P, P, P, I, I, I, P, I, I, I, I, I, ctypes.c_float, I, P]
    t = lib.ce_v_transpose_bf16
    t.restype = I
    t.argtypes = [P, I, P, I, I, I, P]
    return f, t


def main():
    libs = [bind(p) for p in sys.argv[1:]]
    names = [p.split("/")[-1].replace("lib", "").replace(".so", "") for p in sys.argv[1:]]  # NOTE: one knob value per library FILE (the knob is a global of the library)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    for (N, H, B) in [(7200, 40, 2), (13064, 40, 2), (28800, 40, 1)]:
        D = H * 128
        qkv = torch.randn(B * N, 3 * D, generator=g).to(BF).to(dev)
        ldvt = (B * N + 63) // 64 * 64 + 64
        vt = torch.zeros(D, ldvt, dtype=BF, device=dev)
        assert libs[0][1](qkv[:, 2 * D:].data_ptr(), 3 * D, vt.data_ptr(), ldvt, B * N, H, st) == 0
        outs = [torch.empty(B * N, D, dtype=BF, device=dev) for _ in libs]

        def run(f, o):
            rc = f(qkv.data_ptr(), qkv[:, D:].data_ptr(), vt.data_ptr(), N, 3 * D, ldvt, o.data_ptr(), N, H, 128, 3 * D, D, 128 ** -0.5, B, st)
            assert rc == 0, rc

        def timeit(f, o, iters=6):
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
            for i, (f, _) in enumerate(libs):
                best[i] = min(best[i], timeit(f, outs[i]))
        fl = 4.0 * N * N * 128 * H * B
        line = f"attn vt {N}x{N} H{H} B{B}:"
        for i, n in enumerate(names):
            d = (outs[0].float() - outs[i].float()).abs().max().item()
            line += f" | {n} {best[i]:.3f} ms {fl/best[i]/1e9:.0f} TF ({(best[0]/best[i]-1)*100:+.1f} %, d {d:.1e})"
        print(line, flush=True)
        del qkv, outs, vt


if __name__ == "__main__":
    main()
