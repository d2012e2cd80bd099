import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from chronoedit_amd import ops
dev = torch.device("cuda:0"); H = 40; D = H * 128
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3
for rep in range(2):
    for N in (6912, 7200, 7424, 7680, 8192):
        qkv = torch.randn(2 * N, 3 * D, device=dev).to(torch.bfloat16)
        out = torch.empty(2 * N, D, dtype=torch.bfloat16, device=dev)
        t = timeit(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], H, out=out, batch=2))
        nb = 2 * H * ((N + 255) // 256)
        print(f"N={N} blocks={nb} rounds={nb/256:.2f}: {t*1e3:.3f} ms  {2*4.0*N*N*128*H/t/1e12:.1f} TF", flush=True)
        del qkv, out
