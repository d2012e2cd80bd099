"""Timing ablations of the 256x256 LDS-DMA GEMM kernel: one side library per variant (-DCE_GEMM_ABL=n; results are
garbage, durations are the point).  Build here:  python tools/gemm_ablate.py build     Run on the GPU box:  python tools/gemm_ablate.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chronoedit_amd", "csrc")
NAMES = {0: "full kernel", 1: "no barriers in the loop", 2: "no LDS-DMA staging in the loop", 3: "no fragment reads in the loop",
         4: "no MFMAs", 5: "LDS-DMA stream only (no barriers / reads / MFMAs)",
         6: "LDS-DMA stream only, no vmcnt waits", 7: "full kernel, buffer_load..lds instead of global_load_lds"}


def lib_path(a):
    return os.path.join(ROOT, "chronoedit_amd", "lib", f"libgemm_abl{a}.so")


def build():
    for a in NAMES:
        defs = ["-DCE_GEMM_ABL=0", "-DCE_GEMM_BUFFER_DMA"] if a == 7 else [f"-DCE_GEMM_ABL={a}"]
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *defs, "-I", CSRC,
               os.path.join(CSRC, "ce_gemm256.hip"), "-o", lib_path(a)]
        subprocess.check_call(cmd)
        print("built", lib_path(a))


def main():
    import torch
    M, N, K = 14400, 15360, 5120
    pad = int(os.environ.get("LD_PAD", "0"))  # leading-dimension padding of A and W in elements (L2 channel spread probe)
    only = [int(x) for x in os.environ.get("ABLS", "").split(",") if x] or list(NAMES)
    dev = torch.device("cuda:0")
    a = torch.randn(M, K + pad, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K + pad, device=dev) * 0.02).to(torch.bfloat16)
    print("ld pad", pad)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    bias = torch.zeros(N, device=dev)
    P, I = ctypes.c_void_p, ctypes.c_int
    libs = {}
    for ab in only:
        lib = ctypes.CDLL(lib_path(ab))
        lib.ce_gemm256_launch.argtypes = [P, P, P, P, I, P, P, I, I, I, I, I, I, I, I, P]
        libs[ab] = lib
    st = torch.cuda.current_stream().cuda_stream
    fl = 2.0 * M * N * K
    for rep in range(2):
        for ab, lib in libs.items():
            def run():
                rc = lib.ce_gemm256_launch(a.data_ptr(), w.data_ptr(), c.data_ptr(), bias.data_ptr(), 0, None, None, M, N, K, K + pad, K + pad, N, 0, 0, st)
                assert rc == 0, rc
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"rep {rep} abl {ab} {NAMES[ab]:34s}: {ms:.3f} ms  ({fl / ms / 1e9:.0f} TF-equivalent)", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else main()
