#!/bin/bash
# round 6, session ad: the transposed store through the LDS (16-byte stores): parity, the V^T forms timed
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "transposed_store or row_bias" 2>&1 | tail -4 | tee $O/r6ad_pytest.txt
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/r6ad_vt_gemm_ab.txt
import torch, sys
sys.path.insert(0, ".")
from chronoedit_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
for M in (14400, 7200):
    x = torch.randn(M, 5120, generator=g).to(BF).to(dev)
    w = (torch.randn(5120, 5120, generator=g) * 0.02).to(BF).to(dev)
    b = torch.randn(5120, generator=g).to(dev)
    vt = torch.zeros(5120, ops.vt_columns(M), dtype=BF, device=dev)
    o2 = torch.empty(M, 5120, dtype=BF, device=dev)
    def t(fn, iters=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    best = {"row": 1e9, "T": 1e9, "plain": 1e9}
    for _ in range(4):
        best["row"] = min(best["row"], t(lambda: ops.gemm(w, x, b, out=vt[:, :M], epilogue=ops.EPI_BIAS_ROW)))
        best["T"] = min(best["T"], t(lambda: ops.gemm(x, w, b, out=vt[:, :M], epilogue=ops.EPI_BIAS_T)))
        best["plain"] = min(best["plain"], t(lambda: ops.gemm(x, w, b, out=o2)))
    fl = 2.0 * M * 5120 * 5120
    print(f"V^T at M = {M}: swapped operands + row bias {best['row']:.3f} ms {fl / best['row'] / 1e9:.0f} TF | transposed store {best['T']:.3f} ms {fl / best['T'] / 1e9:.0f} TF | "
          f"(the same product stored untransposed: {best['plain']:.3f} ms {fl / best['plain'] / 1e9:.0f} TF)", flush=True)
PY
