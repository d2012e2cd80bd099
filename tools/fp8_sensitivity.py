"""What fp8 costs in accuracy at the FULL width, Linear by Linear: one block of the 14B width (D = 5120, 40 heads, F = 13 824,
512 + 257 context rows) at the 720p shape (N = 7 200) against the fp32 oracle - everything in bf16, ONE of the six large Linears on the MX fp8 GEMM at a
time, only the MXFP8 self-attention, the policies of ChronoEditTransformer3DModel.FP8_POLICIES.  The measurement lives in the test suite (the oracle is
test infrastructure: tests/test_bench_shapes_gpu.py::test_full_width_block_bf16_and_fp8_modes_vs_fp32_oracle); this wrapper runs it and prints its table.
profiles/r06_fp8_sensitivity.txt is the round-6 run of the stand-alone form of this tool (same numbers, plus pairs and the block's launch times).
    python tools/fp8_sensitivity.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_bench_shapes_gpu.py"), "-q", "-s", "-m", "gpu", "-k",
                    "bf16_and_fp8_modes and 90-160"], cwd=ROOT, capture_output=True, text=True)
for ln in r.stdout.splitlines():
    if "full-width block" in ln or "fp8 sensitivity" in ln or "quadrature" in ln or "passed" in ln or "failed" in ln:
        print(ln)
sys.exit(r.returncode)
