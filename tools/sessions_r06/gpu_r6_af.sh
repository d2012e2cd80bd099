#!/bin/bash
# round 6, session af: rmsnorm_rope_mxfp8 with its weight / cos-sin loads one chunk ahead and dword scale stores: parity, time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_hip_kernels.py -x -q -m gpu -k "rmsnorm or rope or mxfp8_attention or edit_end" 2>&1 | tail -3 | tee $O/r6af_pytest.txt
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/r6af_rowpass_time.txt
import torch, sys
sys.path.insert(0, ".")
from chronoedit_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
M, D = 14400, 5120
qkv = torch.randn(M, 3 * D, generator=g).to(BF).to(dev)
one = torch.ones(D, device=dev)
cs = torch.randn(7200, 64, 2, generator=g).to(dev)
def t(fn, iters=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
q8, sq = ops.rmsnorm_rope_mxfp8(qkv[:, :D], one, cs, 128, 1e-6, post_scale=ops.MXFP8_Q_SCALE)
us = min(t(lambda: ops.rmsnorm_rope_mxfp8(qkv[:, :D], one, cs, 128, 1e-6, post_scale=ops.MXFP8_Q_SCALE, out=q8, scale=sq)) for _ in range(3))
print(f"rmsnorm_rope_mxfp8 {M} x {D}: {us:.1f} us  {(M * D * 3 + M * D / 32) / us / 1e6:.2f} TB/s (read bf16 + write e4m3 + scales)")
x2 = qkv[:, :2 * D].clone()
us2 = min(t(lambda: ops.rmsnorm_rope_(x2[:, :D], one, cs, 128, 1e-6, x2=x2[:, D:], w2=one)) for _ in range(3))
print(f"rmsnorm_rope (bf16, q and k in one launch) {M} x {D} x 2: {us2:.1f} us  {M * D * 8 / us2 / 1e6:.2f} TB/s")
PY
