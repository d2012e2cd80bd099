"""A/B timing of BUILDS of the MX fp8 GEMM (ce_gemm_mxfp8 / ce_gemm_mxfp8_gelu_quant, csrc/ce_gemm_fp8w4.hip) in ONE process:
    python tools/gemm_mxfp8_ab.py <base.so> [<variant.so> ...]
Each library is a full libchronoedit_hip build (variants: `-DF8_DMA_SCHED=n` on ce_gemm_fp8w4.hip); all run the fp8 step's five GEMM shapes on
the same quantised operands, interleaved, best and median of the rounds; outputs compared with the first library's (bit-identical expected:
the schedule of the LDS-DMA pieces does not touch the arithmetic).  Every library gets its own split-K scratch."""
import ctypes
import statistics
import sys

import torch

BF = torch.bfloat16


def bind(path):
    lib = ctypes.CDLL(path)
    P, I = ctypes.c_void_p, ctypes.c_int
    g = lib.ce_gemm_mxfp8
    g.restype = I
    g.argtypes = [P, P, P, P, P, P, I, P, P] + [I] * 8 + [P]
    gq = lib.ce_gemm_mxfp8_gelu_quant
    gq.restype = I
    gq.argtypes = [P, P, P, P, P, P, P] + [I] * 6 + [P]
    q = lib.ce_quant_rows_mxfp8
    q.restype = I
    q.argtypes = [P, P, P, I, I, I, I, P]
    ws = lib.ce_set_gemm_workspace
    ws.restype = I
    ws.argtypes = [P, ctypes.c_size_t]
    # round 6: the register-direct epilogue takes the W scales in the W order (ce_quant_rows_mxfp8_w); a -DF8_EPI_LDS=1 build (ce_build_info bit 2)
    # or a library from before round 6 (no ce_build_info / no _w entry) takes them in the A order
    lib.w_order = hasattr(lib, "ce_quant_rows_mxfp8_w") and not (lib.ce_build_info() & 4)
    if lib.w_order:
        lib.ce_quant_rows_mxfp8_w.restype = I
        lib.ce_quant_rows_mxfp8_w.argtypes = q.argtypes
    return lib, g, gq, q, ws


def scale_bytes(rows, K):
    return (rows + 127) // 128 * (K // 128) * 512


def main():
    # --cold: every launch takes ANOTHER copy of the weight operand (as many copies as it takes to exceed the 256 MB Infinity Cache twice), so W streams
    # from HBM as it does inside the step, where 16 GB of fp8 weights pass per forward; without it the ten back-to-back launches of a timing loop
    # find W in the Infinity Cache and the LDS-DMA pieces land sooner than they ever do in the model (round 5: +6..8 % stand-alone, +0.3..1.5 % in the step)
    cold = "--cold" in sys.argv
    # --sustain: 300 launches per timing instead of 10 (100-250 ms of uninterrupted matrix work: the package reaches its power limit as it does inside
    # the 225 ms step; a 5 ms burst between idle gaps runs at a clock the step never sees)
    iters_default = 300 if "--sustain" in sys.argv else 10
    paths = [a for a in sys.argv[1:] if not a.startswith("--")]
    libs = [bind(p) for p in paths]
    names = [p.split("/")[-1].replace("lib", "").replace(".so", "") for p in paths]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    scratch = [torch.empty(96 << 20, dtype=torch.uint8, device=dev) for _ in libs]
    for (_, _, _, _, ws), buf in zip(libs, scratch):
        assert ws(buf.data_ptr(), buf.numel()) == 0
    u8 = lambda *s: torch.empty(s, dtype=torch.uint8, device=dev)
    total = [0.0] * len(libs)
    shapes = [("q|k|v", 14400, 15360, 5120, 0, 1), ("cross q", 14400, 5120, 5120, 0, 1), ("out-proj x2", 14400, 5120, 5120, 2, 2),
              ("ffn-up+gelu+quant", 14400, 13824, 5120, 7, 1), ("ffn-down", 14400, 5120, 13824, 2, 1), ("ffn-up 1584x1056", 26136, 13824, 5120, 7, 0)]
    for (what, M, N, K, epi, per_block) in shapes:
        a = torch.randn(M, K, generator=g).to(BF).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(dev)
        aq, wq, sa, sw = u8(M, K), u8(N, K), u8(scale_bytes(M, K)), u8(scale_bytes(N, K))
        assert libs[0][3](a.data_ptr(), aq.data_ptr(), sa.data_ptr(), M, K, K, K, st) == 0
        assert libs[0][3](w.data_ptr(), wq.data_ptr(), sw.data_ptr(), N, K, K, K, st) == 0
        sw_w = u8(scale_bytes(N, K))  # the same scales in the W order, for the libraries that want them there
        for L in libs:
            if L[0].w_order:
                assert L[0].ce_quant_rows_mxfp8_w(w.data_ptr(), wq.data_ptr(), sw_w.data_ptr(), N, K, K, K, st) == 0
                break
        ncopy = max(1, -(-(600 << 20) // (N * K))) if cold else 1
        wqs, sws = [wq] + [wq.clone() for _ in range(ncopy - 1)], [sw] + [sw.clone() for _ in range(ncopy - 1)]
        sws_w = [sw_w] + [sw_w.clone() for _ in range(ncopy - 1)]
        turn = [0]
        b = torch.randn(N, generator=g).to(dev)
        gate = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(BF).to(dev) if epi == 2 else None
        if epi == 7:
            outs = [u8(M, N) for _ in libs]
            osc = [u8(scale_bytes(M, N)) for _ in libs]
        else:
            outs = [torch.empty(M, N, dtype=BF, device=dev) for _ in libs]

        def run(i):
            _, f, fq, _, _ = libs[i]
            turn[0] = (turn[0] + 1) % ncopy
            wq_, sw_ = wqs[turn[0]], (sws_w if libs[i][0].w_order else sws)[turn[0]]
            if epi == 7:
                rc = fq(aq.data_ptr(), wq_.data_ptr(), sa.data_ptr(), sw_.data_ptr(), b.data_ptr(), outs[i].data_ptr(), osc[i].data_ptr(), M, N, K, K, K, N, st)
            else:
                rc = f(aq.data_ptr(), wq_.data_ptr(), outs[i].data_ptr(), sa.data_ptr(), sw_.data_ptr(), b.data_ptr(), epi, gate.data_ptr() if epi == 2 else None,
                       res.data_ptr() if epi == 2 else None, M, N, K, K, K, N, N, 0, st)
            assert rc == 0, rc

        def timeit(i, iters=iters_default):
            run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters

        ts = [[] for _ in libs]
        for _ in range(3 if iters_default > 10 else 5):
            for i in range(len(libs)):
                ts[i].append(timeit(i))
        fl = 2.0 * M * N * K
        line = f"{what:18s} {M}x{N}x{K} epi{epi}:"
        for i, n in enumerate(names):
            best, med = min(ts[i]), statistics.median(ts[i])
            same = torch.equal(outs[0], outs[i])
            total[i] += per_block * med
            line += f" | {n} best {best:.3f} ms {fl/best/1e9:.0f} TF, median {fl/med/1e9:.0f} TF ({(statistics.median(ts[0])/med-1)*100:+.1f} %{'' if same else ', DIFFERS'})"
        print(line, flush=True)
        del a, w, res, outs, aq, wq, wqs, sws
    print(("COLD weights (HBM) - " if cold else "weights hot in the Infinity Cache - ") + "sum over one block's large fp8 GEMMs, 720p pair (median ms):", {n: round(t, 3) for n, t in zip(names, total)})


if __name__ == "__main__":
    main()
