"""Can RCCL collectives (torch.distributed backend "nccl") be captured into a hipGraph here?  One rank (the only RCCL group a one-GPU
box can form), stage by stage, each stage announced before it runs so that a hang names itself:
    python tools/rccl_graph_probe.py [max_stage]"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stage_max = int(sys.argv[1]) if len(sys.argv) > 1 else 9
MODE = sys.argv[2] if len(sys.argv) > 2 else "global"  # capture_error_mode of torch.cuda.graph


def say(*a):
    print(*a, flush=True)


os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29577"), HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
a = torch.arange(1 << 20, device=dev, dtype=torch.bfloat16).view(1, -1)
b = torch.empty_like(a)
say("capture_error_mode =", MODE)
say("stage 1: eager all_to_all_single")
dist.all_to_all_single(b, a)
torch.cuda.synchronize()
say("  ok", bool(torch.equal(a, b)))
if stage_max >= 2:
    say("stage 2: capture a synchronous all_to_all_single")
    g = torch.cuda.CUDAGraph()
    b.zero_()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dist.all_to_all_single(b, a)  # warm-up on the side stream, as torch recommends before capture
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g, capture_error_mode=MODE):
        dist.all_to_all_single(b, a)
    b.zero_()
    g.replay()
    torch.cuda.synchronize()
    say("  replay ok", bool(torch.equal(a, b)))
if stage_max >= 3:
    say("stage 3: fork / join on a side stream (synchronous collective there, a kernel on the main stream meanwhile)")
    g2 = torch.cuda.CUDAGraph()
    c = torch.empty_like(a)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.graph(g2, capture_error_mode=MODE):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dist.all_to_all_single(b, a)
        c.copy_(a).mul_(2)
        torch.cuda.current_stream().wait_stream(side)
        c.add_(b)
    b.zero_()
    g2.replay()
    torch.cuda.synchronize()
    say("  replay ok", bool(torch.equal(c, a * 2 + a)))
if stage_max >= 6:
    say("stage 6: async_op=True + work.wait() under capture (known to crash this stack: run last, on its own)")
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4, capture_error_mode=MODE):
        w = dist.all_to_all_single(b, a, async_op=True)
        w.wait()
    g4.replay()
    torch.cuda.synchronize()
    say("  replay ok")
if stage_max >= 4:
    say("stage 4: all_gather_into_tensor captured")
    g3 = torch.cuda.CUDAGraph()
    o = torch.empty_like(a)
    with torch.cuda.graph(g3, capture_error_mode=MODE):
        dist.all_gather_into_tensor(o, a)
    o.zero_()
    g3.replay()
    torch.cuda.synchronize()
    say("  replay ok", bool(torch.equal(o, a)))
if stage_max >= 5:
    say("stage 5: the engine's sharded denoising step (GraphedDenoiser over Ulysses(force=True))")
    from chronoedit_amd.pipeline import denoise
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    from oracle import dit_oracle as O
    BF = torch.bfloat16
    cfg = O.DiTConfig(num_attention_heads=4, ffn_dim=1024, num_layers=2, text_dim=128, image_dim=64, added_kv_proj_dim=512)
    m = ChronoEditTransformer3DModel(num_attention_heads=4, in_channels=36, ffn_dim=1024, num_layers=2, text_dim=128, image_dim=64, added_kv_proj_dim=512, device="cuda:0")
    m.load_synthetic_({k: v.cuda() for k, v in O.make_synthetic_params(cfg, dtype=BF).items()})
    m.enable_sequence_parallel(force=True)
    gg = torch.Generator().manual_seed(11)
    lat0 = torch.randn(1, 16, 2, 8, 12, generator=gg).cuda()
    cond = torch.randn(1, 20, 2, 8, 12, generator=gg).cuda().to(BF)
    pr, ng = torch.randn(1, 40, 128, generator=gg).cuda().to(BF), torch.randn(1, 40, 128, generator=gg).cuda().to(BF)
    img = torch.randn(1, 257, 64, generator=gg).cuda().to(BF)
    eager = denoise(m, FlowUniPCMultistepScheduler(flow_shift=5.0), lat0.clone(), cond, pr, ng, img, 3, 5.0).clone()
    say("  eager loop done")
    graphed = denoise(m, FlowUniPCMultistepScheduler(flow_shift=5.0), lat0.clone(), cond, pr, ng, img, 3, 5.0, use_graph=True)
    torch.cuda.synchronize()
    say("  graphed loop done, bit-equal:", bool(torch.equal(eager, graphed)))
dist.destroy_process_group()
say("done")
