"""A/B of the loop BODIES of the V^T attention kernel in one process: `lib[@knob]` arguments, each a build of ce_attn.hip (hipcc -shared) with
the ce_set_attention_waves value to run it under (0 = the 8-wave software-pipelined body, 128 / 129 = one wave per SIMD, one workgroup per
item / persistent).  Interleaved, best of each; outputs compared with the first.
    python tools/attn_body_ab.py chronoedit_amd/lib/libattn_r3.so@0 chronoedit_amd/lib/libchronoedit_hip.so@0 chronoedit_amd/lib/libchronoedit_hip.so@128 ..."""
import ctypes
import sys

import torch

BF = torch.bfloat16
P, I = ctypes.c_void_p, ctypes.c_int
_libs = {}


def bind(spec):
    path, _, knob = spec.partition("@")
    lib = _libs.get(path)
    if lib is None:
        lib = _libs[path] = ctypes.CDLL(path)
        lib.ce_attention_vt_bf16.restype = I
        lib.ce_attention_vt_bf16.argtypes = [P, P, P, I, I, I, P, I, I, I, I, I, ctypes.c_float, I, P]
        lib.ce_v_transpose_bf16.restype = I
        lib.ce_v_transpose_bf16.argtypes = [P, I, P, I, I, I, P]
        lib.ce_set_attention_waves.restype = I
        lib.ce_set_attention_waves.argtypes = [I]
    return lib, int(knob or 0), spec.split("/")[-1].replace("lib", "").replace(".so", "")


def main():
    variants = [bind(a) for a in sys.argv[1:]]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    for (N, H, B) in [(7200, 40, 2), (28800, 40, 1), (28800, 5, 2), (13068, 40, 2)]:
        D = H * 128
        qkv = torch.randn(B * N, 3 * D, generator=g).to(BF).to(dev)
        ldvt = (B * N + 63) // 64 * 64 + 64
        vt = torch.zeros(D, ldvt, dtype=BF, device=dev)
        assert variants[0][0].ce_v_transpose_bf16(qkv[:, 2 * D:].data_ptr(), 3 * D, vt.data_ptr(), ldvt, B * N, H, st) == 0
        outs = [torch.empty(B * N, D, dtype=BF, device=dev) for _ in variants]

        def run(v, o):
            lib, knob, _ = v
            lib.ce_set_attention_waves(knob)
            rc = lib.ce_attention_vt_bf16(qkv.data_ptr(), qkv[:, D:].data_ptr(), vt.data_ptr(), N, 3 * D, ldvt, o.data_ptr(), N, H, 128, 3 * D, D,
                                          128 ** -0.5, B, st)
            assert rc == 0, rc

        def timeit(v, o, iters=6):
            run(v, o)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run(v, o)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters

        best = [1e9] * len(variants)
        for _ in range(4):
            for i, v in enumerate(variants):
                best[i] = min(best[i], timeit(v, outs[i]))
        fl = 4.0 * N * N * 128 * H * B
        line = f"attn vt {N}x{N} H{H} B{B}:"
        for i, v in enumerate(variants):
            d = (outs[0].float() - outs[i].float()).abs().max().item()
            line += f" | {v[2]} {best[i]:.3f} ms {fl / best[i] / 1e9:.0f} TF ({(best[0] / best[i] - 1) * 100:+.1f} %, d {d:.1e})"
        print(line, flush=True)
        del qkv, outs, vt


if __name__ == "__main__":
    main()
