"""Timing ablations of the ping-pong attention kernel (drop one ingredient of the loop at a time; results are garbage,
only the durations mean something).  Build here (no GPU needed):  python tools/attn_ablate.py build
Run on the GPU box:                                               python tools/attn_ablate.py [Nq]
The side library (ce_attn.hip with -DCE_ATTN_ABLATE) never replaces the product library."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chronoedit_amd", "csrc")
LIB = os.path.join(ROOT, "chronoedit_amd", "lib", "libattn_ablate.so")
NAMES = {0: "full kernel", 1: "no exp2 (fma only)", 2: "no P.V MFMAs", 3: "no K.Q^T MFMAs", 4: "no global loads / LDS stores",
         5: "no barriers in the loop", 6: "prio 3 on the exp segment only", 7: "prio 3 on all of phase 2",
         8: "static prio 1 for group 1", 9: "prio 3 on P.V MFMAs, 0 on exp"}
if os.environ.get("ATTN_KERNEL") == "64":
    NAMES = {0: "full kernel", 1: "no exp2 (fma only)", 2: "no P.V MFMAs", 3: "no K.Q^T MFMAs", 4: "no staging (stores + fetches)",
             7: "no global fetches (stores kept)", 8: "no LDS stores (fetches kept)",
             11: "no K LDS-DMA (V fetches kept)", 12: "no V fetches (K DMA kept)"}


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    extra = os.environ.get("ATTN_DEFS", "").split()
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCE_ATTN_ABLATE", *extra, "-I", CSRC,
           os.path.join(CSRC, "ce_attn.hip"), "-o", LIB]
    subprocess.check_call(cmd)
    print("built", LIB)


def main():
    import torch
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 7200
    H, D = 40, 40 * 128
    lib = ctypes.CDLL(LIB)
    dev = torch.device("cuda:0")
    qkv = torch.randn(N, 3 * D, device=dev).to(torch.bfloat16)
    out = torch.empty(N, D, dtype=torch.bfloat16, device=dev)
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.ce_attention_bf16.argtypes = [P, P, P, I, I, I, P, P, I, I, I, P, I, I, I, I, I, ctypes.c_float, P]
    kern = int(os.environ.get("ATTN_KERNEL", "32"))  # 32 ping-pong, 64 software-pipelined
    lib.ce_set_attention_waves(kern)
    print("kernel knob", kern)
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D
    st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = lib.ce_attention_bf16(q, k, v, N, 3 * D, 3 * D, None, None, 0, 0, 0, out.data_ptr(), N, H, 128, 3 * D, D, 128 ** -0.5, st)
        assert rc == 0, rc

    fl = 4.0 * N * N * 128 * H
    ABLS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else list(NAMES)
    for rep in range(2):
        for a in ABLS:
            lib.ce_attn_set_ablation(a)
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
            print(f"rep {rep} abl {a} {NAMES[a]:32s}: {ms:.3f} ms  ({fl / ms / 1e9:.0f} TF-equivalent)", flush=True)


def stamps():
    """ABL 10: s_memtime stamps of workgroup 300, tiles 40..55: [0] phase-1 start (after barrier), [1] phase-1 end (before
    barrier), [2] phase-2 start, [3] end of phase 2 of the PREVIOUS tile (before the barrier that opens this tile)."""
    import numpy as np
    import torch
    N, H, D = 7200, 40, 40 * 128
    lib = ctypes.CDLL(LIB)
    dev = torch.device("cuda:0")
    qkv = torch.randn(N, 3 * D, device=dev).to(torch.bfloat16)
    out = torch.empty(N, D, dtype=torch.bfloat16, device=dev)
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.ce_attention_bf16.argtypes = [P, P, P, I, I, I, P, P, I, I, I, P, I, I, I, I, I, ctypes.c_float, P]
    lib.ce_attn_read_ts.argtypes = [P]
    lib.ce_set_attention_waves(32)
    lib.ce_attn_set_ablation(10)
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D
    for _ in range(3):
        rc = lib.ce_attention_bf16(q, k, v, N, 3 * D, 3 * D, None, None, 0, 0, 0, out.data_ptr(), N, H, 128, 3 * D, D, 128 ** -0.5,
                                   torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    ts = np.zeros((8, 16, 4), dtype=np.uint64)
    assert lib.ce_attn_read_ts(ts.ctypes.data) == 0
    ts = ts.astype(np.int64)
    t0 = ts[:, 1:, :].min()
    print("cycles relative to the first stamp; rows = tiles 41..54; per wave: p1 start | p1 end | p2 start | p2 end(next tile's [3])")
    for w in range(8):
        print(f"wave {w} (group {w >> 2})")
        for t in range(1, 15):
            p1s, p1e, p2s = ts[w, t, 0] - t0, ts[w, t, 1] - t0, ts[w, t, 2] - t0
            p2e = ts[w, t + 1, 3] - t0
            print(f"  tile {40 + t}: p1 {p1s:7d}..{p1e:7d} ({p1e - p1s:5d})  wait {p2s - p1e:5d}  p2 {p2s:7d}..{p2e:7d} ({p2e - p2s:5d})  "
                  f"wait {ts[w, t + 1, 0] - t0 - p2e:5d}")


def stamps_sp():
    """sp kernel, ABL 10: issue-time stamps of workgroup 300, tiles 40..55: [0] after the barrier, [1] stores / fetches issued,
    [2] row max done (S(t) complete), [3] P.V / exp2 section done, [4] pack done, [5] arrival at the next barrier."""
    import numpy as np
    import torch
    N, H, D = 7200, 40, 40 * 128
    lib = ctypes.CDLL(LIB)
    dev = torch.device("cuda:0")
    qkv = torch.randn(N, 3 * D, device=dev).to(torch.bfloat16)
    out = torch.empty(N, D, dtype=torch.bfloat16, device=dev)
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.ce_attention_bf16.argtypes = [P, P, P, I, I, I, P, P, I, I, I, P, I, I, I, I, I, ctypes.c_float, P]
    lib.ce_attn_read_ts6.argtypes = [P]
    lib.ce_set_attention_waves(64)
    lib.ce_attn_set_ablation(10)
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D
    for _ in range(3):
        rc = lib.ce_attention_bf16(q, k, v, N, 3 * D, 3 * D, None, None, 0, 0, 0, out.data_ptr(), N, H, 128, 3 * D, D, 128 ** -0.5,
                                   torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    ts = np.zeros((8, 16, 6), dtype=np.uint64)
    assert lib.ce_attn_read_ts6(ts.ctypes.data) == 0
    ts = ts.astype(np.int64)
    t0 = ts[:, 1:, 0].min()
    print("per tile: start | stage(stores+fetch issue) | QK+rowmax | PV+exp | pack | barrier wait | total")
    for w in range(8):
        print(f"wave {w}")
        for t in range(1, 15):
            a = ts[w, t]
            nxt = ts[w, t + 1]
            print(f"  tile {40 + t}: start {a[0] - t0:7d}  stage {a[1] - a[0]:5d}  qk+max {a[2] - a[1]:5d}  pv+exp {a[3] - a[2]:5d}  "
                  f"pack {a[4] - a[3]:5d}  to-barrier {nxt[5] - a[4]:5d}  wait {nxt[0] - nxt[5]:5d}  total {nxt[0] - a[0]:5d}")


def stamps_w4():
    """w4 kernel, ABL 10: stamps of workgroup 300, tiles 40..55 at the section boundaries."""
    import numpy as np
    import torch
    N, H, D = 7200, 40, 40 * 128
    lib = ctypes.CDLL(LIB)
    dev = torch.device("cuda:0")
    qkv = torch.randn(N, 3 * D, device=dev).to(torch.bfloat16)
    out = torch.empty(N, D, dtype=torch.bfloat16, device=dev)
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.ce_attention_bf16.argtypes = [P, P, P, I, I, I, P, P, I, I, I, P, I, I, I, I, I, ctypes.c_float, P]
    lib.ce_attn_read_ts6.argtypes = [P]
    lib.ce_set_attention_waves(128)
    lib.ce_attn_set_ablation(10)
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D
    for _ in range(3):
        rc = lib.ce_attention_bf16(q, k, v, N, 3 * D, 3 * D, None, None, 0, 0, 0, out.data_ptr(), N, H, 128, 3 * D, D, 128 ** -0.5,
                                   torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    ts = np.zeros((8, 16, 6), dtype=np.uint64)
    assert lib.ce_attn_read_ts6(ts.ctypes.data) == 0
    ts = ts.astype(np.int64)
    for w in range(4):
        print(f"wave {w}: S_A|smB  PV_B|stage  S_B|smA  barrier  PV_A  total")
        for t in range(1, 15):
            a = ts[w, t]
            print(f"  tile {40 + t}: {a[1] - a[0]:6d} {a[2] - a[1]:6d} {a[3] - a[2]:6d} {a[4] - a[3]:6d} {a[5] - a[4]:6d}   {ts[w, t + 1, 0] - a[0]:6d}")


if __name__ == "__main__":
    if sys.argv[1:2] == ["stamps_w4"]:
        stamps_w4()
        sys.exit(0)
    if sys.argv[1:2] == ["stamps_sp"]:
        stamps_sp()
        sys.exit(0)
    if sys.argv[1:2] == ["stamps"]:
        stamps()
        sys.exit(0)
    build() if sys.argv[1:2] == ["build"] else main()
