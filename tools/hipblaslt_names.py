"""Which hipBLASLt kernels does torch.matmul pick for the step's GEMM shapes?  (yardstick only; run under rocprofv3 --kernel-trace)"""
import torch
dev = torch.device("cuda:0")
for (M, N, K) in [(14400, 15360, 5120), (14400, 13824, 5120), (14400, 5120, 13824), (14400, 5120, 5120), (28800, 13824, 5120), (512, 5120, 5120)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    for _ in range(3):
        c = torch.matmul(a, w.t())
    torch.cuda.synchronize()
