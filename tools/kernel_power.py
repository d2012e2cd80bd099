"""Energy per flop of the step's kernels: each kernel launched back to back for ~3 s while rocm-smi is read twice (package power, shader clock), its rate from
HIP events over the same window -> W / (TFLOP/s) = pJ per flop.  The chip is power-limited inside the step (profiles/r05_power_clock_trace.txt), so THIS is the
figure of merit that decides the step time; 249 W of the reading is the idle floor.    python tools/kernel_power.py"""
import os
import re
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator().manual_seed(0)


def smi():
    txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
    pw = re.search(r"Package Power \(W\): ([0-9.]+)", txt)
    ck = re.search(r"sclk clock level: \w+: \((\d+)Mhz\)", txt)
    return (float(pw.group(1)) if pw else float("nan")), (int(ck.group(1)) if ck else -1)


def measure(name, fn, flops, bytes_=0, seconds=3.0):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    n = max(20, int(seconds * 1e3 / max(e0.elapsed_time(e1), 1e-3)))
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    time.sleep(seconds * 0.4)
    p1, c1 = smi()
    time.sleep(seconds * 0.25)
    p2, c2 = smi()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    pw, ck = (p1 + p2) / 2, (c1 + c2) / 2
    if flops:
        tf = flops / ms / 1e9
        print(f"{name:58s} {ms:8.3f} ms {tf:7.0f} TFLOP/s | {pw:6.0f} W {ck:5.0f} MHz | {pw / tf:5.2f} pJ/flop (dynamic, above the 249 W idle floor: {(pw - 249) / tf:5.2f})", flush=True)
    else:
        print(f"{name:58s} {ms:8.3f} ms {bytes_ / ms / 1e9:7.2f} TB/s    | {pw:6.0f} W {ck:5.0f} MHz | {pw / (bytes_ / ms / 1e9) :5.1f} pJ/byte", flush=True)
    time.sleep(1.0)


M, D, F, H = 14400, 5120, 13824, 40
a = torch.randn(M, D, generator=g).to(BF).to(dev)
af = torch.randn(M, F, generator=g).to(BF).to(dev)
w_up = (torch.randn(F, D, generator=g) * 0.02).to(BF).to(dev)
w_dn = (torch.randn(D, F, generator=g) * 0.02).to(BF).to(dev)
w_o = (torch.randn(D, D, generator=g) * 0.02).to(BF).to(dev)
b_up, b_d = torch.zeros(F, device=dev), torch.zeros(D, device=dev)
gate = torch.ones(D, device=dev)
o_up = torch.empty(M, F, dtype=BF, device=dev)
o_d = torch.zeros(M, D, dtype=BF, device=dev)
print("idle:", smi())
measure("bf16 GEMM FFN-up 14400x13824x5120 (bias + GELU)", lambda: ops.gemm(a, w_up, b_up, out=o_up, epilogue=ops.EPI_BIAS_GELU), 2.0 * M * F * D)
measure("bf16 GEMM FFN-down 14400x5120x13824 (gated residual)", lambda: ops.gemm(af, w_dn, b_d, out=o_d, epilogue=ops.EPI_GATE_RES, gate=gate, res=o_d), 2.0 * M * F * D)
measure("bf16 GEMM out-projection 14400x5120x5120 (gated residual)", lambda: ops.gemm(a, w_o, b_d, out=o_d, epilogue=ops.EPI_GATE_RES, gate=gate, res=o_d), 2.0 * M * D * D)
qkv = torch.randn(M, 3 * D, generator=g).to(BF).to(dev)
vt = ops.v_transpose(qkv[:, 2 * D:], H)
o_att = torch.empty(M, D, dtype=BF, device=dev)
measure("bf16 self-attention 7200 keys x 40 heads x 2 (V^T form)", lambda: ops.attention_vt(qkv[:, :D], qkv[:, D:2 * D], vt, H, out=o_att, batch=2), 4.0 * 7200 * 7200 * 128 * H * 2)
aq, sa = ops.quant_rows_mxfp8(a)
wq, sw = ops.quant_rows_mxfp8(w_up, w_order=True)
oq = torch.empty(M, F, dtype=torch.uint8, device=dev)
so = torch.empty(ops.mx_scale_bytes(M, F), dtype=torch.uint8, device=dev)
measure("MX fp8 GEMM FFN-up (bias + GELU + MX quantiser)", lambda: ops.gemm_mxfp8_gelu_quant(aq, sa, wq, sw, b_up, oq, so), 2.0 * M * F * D)
wdq, swd = ops.quant_rows_mxfp8(w_dn, w_order=True)
measure("MX fp8 GEMM FFN-down (gated residual)", lambda: ops.gemm_mxfp8(oq, so, wdq, swd, b_d, out=o_d, epilogue=ops.EPI_GATE_RES, gate=gate, res=o_d), 2.0 * M * F * D)
one = torch.ones(D, device=dev)
q8, sq = ops.rmsnorm_rope_mxfp8(qkv[:, :D], one, None, 128, 1e-6, post_scale=ops.MXFP8_Q_SCALE)
k8, sk = ops.rmsnorm_rope_mxfp8(qkv[:, D:2 * D], one, None, 128, 1e-6)
v8t, sv = ops.v_mxfp8_transpose(qkv[:, 2 * D:], 7200, 2, H)
measure("MXFP8 self-attention 7200 keys x 40 heads x 2", lambda: ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, out=o_att, batch=2), 4.0 * 7200 * 7200 * 128 * H * 2)
ones, zeros = torch.ones(D, device=dev), torch.zeros(D, device=dev)
h = torch.empty_like(a)
measure("LN-modulate row pass 14400x5120 (HBM-bound)", lambda: ops.ln_affine(a, ones, zeros, 1e-6, out=h), 0, bytes_=4.0 * M * D)
