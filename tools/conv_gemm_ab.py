"""The VAE's wide stride-1 convs at the 720p decode shapes: ce_conv3d_gemm_bf16 with each macro tile (256 x 96, 256 x 128, 256 x 256) vs the
implicit-GEMM kernel (ce_conv_igemm_bf16), one process, interleaved; every tile's output is compared with the 128-wide one's (same products,
same summation order per accumulator: bit-identical; kind 1 = the slab kernel of the 96-channel layers: another summation order).   python tools/conv_gemm_ab.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402
from chronoedit_amd.vae import Frames, _ConvPack  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    shapes = [(3, 192, 192, 4, 360, 640), (3, 384, 384, 2, 180, 320), (3, 384, 384, 1, 90, 160), (3, 96, 96, 4, 720, 1280),
              (3, 96, 96, 1, 720, 1280), (1, 384, 192, 4, 360, 640), (1, 192, 96, 4, 720, 1280), (3, 192, 384, 2, 180, 320)]
    for (KT, Cin, Cout, T, H, W) in shapes:
        n_in = T + KT - 1
        f = Frames(T, H, W, Cin, dev, front=KT - 1)
        f.stack[:n_in, 1:-1, 1:-1] = torch.randn(n_in, H, W, Cin, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(Cout, Cin, KT, 3, 3, generator=g) / (9 * KT * Cin) ** 0.5).to(torch.bfloat16).to(dev)
        pk = _ConvPack(w, torch.randn(Cout, generator=g).to(dev))
        out = Frames(T, H, W, Cout, dev)
        fl = 2.0 * T * H * W * Cout * Cin * KT * 9

        def run(kind):
            if kind == "old":
                ops.conv_igemm([f.stack[i] for i in range(n_in)], pk.w, pk.b, out.frame_list(), None, Cin=Cin, Cout=Cout, KT=KT, KH=3, KW=3,
                               st=1, ss=1, H_out=H, W_out=W, in_Wp=W + 2, in_off=0, out_Wp=W + 2, out_border=1, out_cstride=Cout)
            else:
                ops.conv3d_gemm(f.stack, pk.gemm_weight(), pk.b, out.data, None, T_out=T, H=H, W=W, Cin=Cin, Cout=Cout, KT=KT, n_tile=kind)

        def timeit(kind, iters=5):
            run(kind)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run(kind)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters

        best = {}
        kinds = ("old", 96, 128, 256) if Cout % 96 == 0 else ("old", 128, 256)
        if Cout == 96 and Cin in (96, 192):
            kinds = kinds + (1, 2)  # round 6: the input slab in the LDS (conv3x3_c96_*kernel), 510 positions x 96 channels per workgroup: 8 | 4 waves
        if os.environ.get("CE_CONV_AB_ONLY96") == "1" and Cout != 96:
            continue
        for _ in range(rounds):
            for kind in kinds:
                best[kind] = min(best.get(kind, 1e9), timeit(kind))
        run(128)
        ref = out.data.clone()
        same = {}
        for kind in kinds[1:]:
            out.data.zero_()
            run(kind)
            same[kind] = bool(torch.equal(out.data, ref)) if kind not in (1, 2) else f"rel-L2 {float((out.data.float() - ref.float()).norm() / ref.float().norm()):.1e}"
        print(f"conv {KT}x3x3 {Cin}->{Cout} {T}x{H}x{W}: " + " | ".join(f"{k}: {v:.3f} ms {fl / v / 1e9:.0f} TF" for k, v in best.items()) +
              f" | == 128-wide: {same}", flush=True)
        del f, out


if __name__ == "__main__":
    main()
