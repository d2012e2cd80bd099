"""VAE mid-block attention (ce_attention_1head_bf16): time and error with and without the key split, at the frame sizes of the engine."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for N, C in ((14400, 384), (3600, 384), (57600 // 4 * 3, 384), (26136, 384)):
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(N, 3 * C, generator=g).to(torch.bfloat16).to(dev)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    hwp = (N + 63) // 64 * 64
    vt = torch.zeros((C, hwp), dtype=torch.bfloat16, device=dev)
    vt[:, :N] = v.t()
    ref = torch.softmax(q[:1024].float() @ k.float().t() * C ** -0.5, dim=-1) @ v.float()
    outs = {}
    for split in (False, True):
        out = ops.attention_1head(q, k, vt, C ** -0.5, split_keys=split)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attention_1head(q, k, vt, C ** -0.5, out=out, split_keys=split)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        err = float((out[:1024].float() - ref).norm() / ref.norm())
        outs[split] = out
        print(f"N {N:6d} C {C} split_keys {split!s:5}: {ms:7.3f} ms  {4.0 * N * N * C / ms / 1e9:7.1f} TFLOP/s  rel-L2 vs fp32 {err:.2e}", flush=True)
    d = float((outs[True].float() - outs[False].float()).norm() / outs[False].float().norm())
    print(f"   split vs unsplit rel-L2 {d:.2e}", flush=True)
