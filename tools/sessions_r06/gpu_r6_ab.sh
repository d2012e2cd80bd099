#!/bin/bash
# round 6, session ab: V^T by a transposed-store epilogue of the 384- / 288-row GEMM (CE_EPI_BIAS_T): parity, the two V^T forms timed, the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "transposed_store or row_bias" 2>&1 | tail -4 | tee $O/r6ab_pytest.txt
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/r6ab_vt_gemm_ab.txt
import torch, sys
sys.path.insert(0, ".")
from chronoedit_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
for M in (14400, 7200, 13068, 26136):
    x = torch.randn(M, 5120, generator=g).to(BF).to(dev)
    w = (torch.randn(5120, 5120, generator=g) * 0.02).to(BF).to(dev)
    b = torch.randn(5120, generator=g).to(dev)
    vt = torch.zeros(5120, ops.vt_columns(M), dtype=BF, device=dev)
    def t(fn, iters=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    best = {"row": 1e9, "T": 1e9}
    for _ in range(4):
        best["row"] = min(best["row"], t(lambda: ops.gemm(w, x, b, out=vt[:, :M], epilogue=ops.EPI_BIAS_ROW)))
        a = vt.clone()
        best["T"] = min(best["T"], t(lambda: ops.gemm(x, w, b, out=vt[:, :M], epilogue=ops.EPI_BIAS_T)))
        same = bool(torch.equal(a, vt))
    fl = 2.0 * M * 5120 * 5120
    print(f"V^T at M = {M}: swapped operands + row bias {best['row']:.3f} ms {fl / best['row'] / 1e9:.0f} TF | transposed store {best['T']:.3f} ms {fl / best['T'] / 1e9:.0f} TF | bit-equal {same}", flush=True)
PY
timeout 900 python -m pytest tests/test_dit_forward_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3
F="--no-vae --no-encoders --no-fp8-leg --no-fp8-config4 --no-cpu-baseline --no-edit --no-full-edit --no-reasoning-edit"
for a in "--steps 10 --warmup 2" "--guidance 1.0 --steps 16 --warmup 2" "--steps 10 --warmup 2"; do
  timeout 600 python bench.py $F $a 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: (v['avg_ms'], v['tflops']) for k, v in d['kernel_breakdown'].items() if 'epi6' in k or 'epi7' in k})"
done
