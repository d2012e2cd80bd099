"""Where does the time of the MXFP8 attention kernel go?  Side builds of ce_attn_fp8.hip with ONE ingredient of the software-pipelined
loop removed (-DCE_FP8_ABL=n; results are garbage, durations are the point), timed in one process against the full kernel.
    python tools/attn8_ablate.py build      (no GPU needed: writes chronoedit_amd/lib/libattn8_abl<n>.so)
    python tools/attn8_ablate.py [N]        (on the GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "chronoedit_amd", "lib")
NAMES = {0: "full kernel", 1: "no exp2 / e4m3 conversion", 2: "no P.V MFMAs", 3: "no K.Q^T MFMAs", 4: "no fragment reads in the loop",
         5: "no LDS-DMA in the loop", 6: "no row-sum MFMA", 7: "no barrier / counted wait in the loop"}


def build():
    os.makedirs(LIB, exist_ok=True)
    src = os.path.join(ROOT, "chronoedit_amd", "csrc", "ce_attn_fp8.hip")
    procs = []
    for n in NAMES:
        out = os.path.join(LIB, f"libattn8_abl{n}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", f"-DCE_FP8_ABL={n}", "-I",
               os.path.join(ROOT, "chronoedit_amd", "csrc"), src, "-o", out]
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        assert p.wait() == 0


def main():
    import torch
    from chronoedit_amd import ops
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 28800
    H, B = 40, 1
    D = H * 128
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * N, 3 * D, generator=g).to(torch.bfloat16).to(dev)
    one = torch.ones(D, device=dev)
    q8, sq = ops.rmsnorm_rope_mxfp8(qkv[:, :D], one, None, 128, 1e-6, post_scale=ops.MXFP8_Q_SCALE)
    k8, sk = ops.rmsnorm_rope_mxfp8(qkv[:, D:2 * D], one, None, 128, 1e-6)
    v8t, sv = ops.v_mxfp8_transpose(qkv[:, 2 * D:], N, B, H)
    out = torch.empty(B * N, D, dtype=torch.bfloat16, device=dev)
    P, I = ctypes.c_void_p, ctypes.c_int
    st = torch.cuda.current_stream().cuda_stream
    fl = 4.0 * N * N * 128 * H * B
    libs = {}
    for n in NAMES:
        lib = ctypes.CDLL(os.path.join(LIB, f"libattn8_abl{n}.so"))
        lib.ce_attention_mxfp8.argtypes = [P] * 7 + [I] * 9 + [P]
        lib.ce_attention_mxfp8.restype = I
        libs[n] = lib

    def run(lib):
        rc = lib.ce_attention_mxfp8(q8.data_ptr(), sq.data_ptr(), k8.data_ptr(), sk.data_ptr(), v8t.data_ptr(), sv.data_ptr(), out.data_ptr(), N, N,
                                    v8t.shape[-1], H, 128, D, D, D, B, st)
        assert rc == 0, rc

    for rep in range(2):
        for n, lib in libs.items():
            run(lib)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(lib)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 3
            print(f"rep {rep} N={N} abl {n} {NAMES[n]:40s}: {t:.3f} ms  ({fl / t / 1e9:.0f} TF-equivalent)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        main()
