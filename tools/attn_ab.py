"""A/B timing of two builds of the attention kernel in ONE process (box-to-box and thermal drift exceed the effects being
measured): usage  python tools/attn_ab.py <base.so> [<variant.so> ...]   (the in-tree library is always the last entry).
The base library is built from an older ce_attn.hip with the same hipcc line as hiplib.build(); both are called through
the C ABI on the same tensors, interleaved A/B/A/B, best of each."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from chronoedit_amd import hiplib  # noqa: E402

BF = torch.bfloat16


def bind(path):
    knob = None
    if "@" in path:  # <lib.so>@<knob>: select a loop body through ce_set_attention_waves
        path, knob = path.split("@")
    lib = ctypes.CDLL(path)
    if knob is not None:
        lib.ce_set_attention_waves(int(knob))
    f = lib.ce_attention_batched_bf16
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p] + \
        [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    return f


def main():
    paths = sys.argv[1:] + [hiplib.LIB_PATH]
    fns = [bind(p) for p in paths]
    names = [p.split("/")[-1].replace("lib", "").replace(".so", "") for p in paths]
    paths = [p.split("@")[0] for p in paths]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for (Nq, Nkv, H, B) in [(7200, 7200, 40, 2), (7200, 7200, 40, 1), (7200, 512, 40, 2), (17424, 17424, 40, 1)]:
        D = H * 128
        qkv = torch.randn(B * max(Nq, Nkv), 3 * D, generator=g).to(BF).to(dev)
        q, k, v = qkv[:B * Nq, :D], qkv[:B * Nkv, D:2 * D], qkv[:B * Nkv, 2 * D:]
        outs = [torch.empty(B * Nq, D, dtype=BF, device=dev) for _ in fns]
        st = torch.cuda.current_stream().cuda_stream

        def run(f, o):
            rc = f(q.data_ptr(), k.data_ptr(), v.data_ptr(), Nkv, 3 * D, 3 * D, None, None, 0, 0, 0, o.data_ptr(), Nq, H, 128,
                   3 * D, D, 128 ** -0.5, B, st)
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

        best = [1e9] * len(fns)
        for _ in range(4):
            for i, f in enumerate(fns):
                best[i] = min(best[i], timeit(f, outs[i]))
        fl = 4.0 * Nq * Nkv * 128 * H * B
        line = f"attn {Nq}x{Nkv} H{H} B{B}:"
        for i, n in enumerate(names):
            d = (outs[0].float() - outs[i].float()).abs().max().item()
            line += f" | {n} {best[i]:.3f} ms {fl/best[i]/1e9:.0f} TF ({(best[0]/best[i]-1)*100:+.1f} %, d {d:.1e})"
        print(line, flush=True)
        del qkv, outs


if __name__ == "__main__":
    main()
