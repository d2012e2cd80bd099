"""Stage-by-stage probe of the library-owned RCCL communicator (csrc/ce_comm.hip) on ONE rank: which call returns, which hangs.
    timeout 120 python tools/owned_comm_probe.py [max_stage]"""
import ctypes
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(45, exit=True)  # a hung stage: print where, and leave


def say(*a):
    print(f"[{time.perf_counter() - T0:7.2f}s]", *a, flush=True)


T0 = time.perf_counter()
max_stage = int(sys.argv[1]) if len(sys.argv) > 1 else 99
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
from chronoedit_amd import hiplib  # noqa: E402

lib = hiplib.load()
path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
say("stage 1: ce_comm_load", lib.ce_comm_load(path.encode()))
buf = (ctypes.c_ubyte * 128)()
say("stage 2: ce_comm_unique_id", lib.ce_comm_unique_id(buf))
if max_stage < 3:
    sys.exit(0)
h = ctypes.c_void_p()
x = torch.zeros(4, device="cuda")  # (HIP context up before RCCL)
torch.cuda.synchronize()
say("stage 3: ce_comm_init (world 1) ...")
rc = lib.ce_comm_init(ctypes.byref(h), buf, 0, 1)
say("   ->", rc)
if max_stage < 4 or rc:
    sys.exit(0)
st = torch.cuda.current_stream().cuda_stream
a = torch.arange(1024, device="cuda", dtype=torch.float32)
b = torch.zeros_like(a)
say("stage 4: ce_comm_all_to_all to self (eager) ...")
rc = lib.ce_comm_all_to_all(h, a.data_ptr(), b.data_ptr(), a.numel() * 4, st)
torch.cuda.synchronize()
say("   ->", rc, bool(torch.equal(a, b)))
if max_stage < 5:
    sys.exit(0)
c = torch.zeros_like(a)
say("stage 5: ce_comm_all_gather (eager) ...")
rc = lib.ce_comm_all_gather(h, a.data_ptr(), c.data_ptr(), a.numel() * 4, st)
torch.cuda.synchronize()
say("   ->", rc, bool(torch.equal(a, c)))
if max_stage < 6:
    sys.exit(0)
say("stage 6: capture all_to_all + a kernel, replay x3 ...")
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
b.zero_()
with torch.cuda.graph(g):
    a2 = a * 2
    rc = lib.ce_comm_all_to_all(h, a2.data_ptr(), b.data_ptr(), a.numel() * 4, torch.cuda.current_stream().cuda_stream)
    d = b + 1
say("   captured, rc", rc)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
say("   replayed:", bool(torch.equal(d, a * 2 + 1)))
if max_stage < 7:
    sys.exit(0)
mode = os.environ.get("PROBE_MODE", "side")
if mode == "plain2":
    say("stage 7': a SECOND capture on the capturing stream only ...")
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        a4 = a * 4
        rc = lib.ce_comm_all_to_all(h, a4.data_ptr(), b.data_ptr(), a.numel() * 4, torch.cuda.current_stream().cuda_stream)
        d4 = b + 2
    g3.replay()
    g.replay()
    g3.replay()
    torch.cuda.synchronize()
    say("   replayed:", bool(torch.equal(d4, a * 4 + 2)))
    say("stage 7'': a THIRD capture ...")
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4):
        a5 = a * 5
        rc = lib.ce_comm_all_to_all(h, a5.data_ptr(), b.data_ptr(), a.numel() * 4, torch.cuda.current_stream().cuda_stream)
        rc = lib.ce_comm_all_gather(h, b.data_ptr(), c.data_ptr(), a.numel() * 4, torch.cuda.current_stream().cuda_stream)
        d5 = c + 3
    g4.replay()
    torch.cuda.synchronize()
    say("   replayed:", bool(torch.equal(d5, a * 5 + 3)))
    sys.exit(0)
if mode == "eager8":
    say("stage 8 (eager only): torch.distributed nccl group of one beside it, OwnedComm through parallel.py ...")
    dist.init_process_group("nccl", rank=0, world_size=1)
    from chronoedit_amd.parallel import OwnedComm  # noqa: E402
    oc = OwnedComm()
    say("   OwnedComm up")
    for i in range(3):
        src_i = (a * (i + 1)).view(1, -1).contiguous()  # (kept alive: the exchange runs asynchronously on the side stream)
        w = oc.all_to_all(src_i, b.view(1, -1), async_op=True)
        e = a + 1
        w.wait()
        torch.cuda.synchronize()
        say("   async all_to_all (side stream, eager):", bool(torch.equal(a * (i + 1), b)))
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        w = oc.all_to_all((a * 7).view(1, -1).contiguous(), b.view(1, -1), async_op=True)
        w.wait()
        r7 = b + 1
    gg.replay()
    torch.cuda.synchronize()
    say("   the same call under capture (runs on the capturing stream):", bool(torch.equal(r7, a * 7 + 1)))
    oc.close()
    dist.destroy_process_group()
    say("done (ncclCommDestroy skipped: it blocks on this stack)")
    faulthandler.cancel_dump_traceback_later()
    os._exit(0)
say("stage 7: a second capture with a side-stream fork / join ...")
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    a3 = a * 3
    side.wait_stream(torch.cuda.current_stream())
    rc = lib.ce_comm_all_to_all(h, a3.data_ptr(), b.data_ptr(), a.numel() * 4, side.cuda_stream)
    ev = torch.cuda.Event()
    ev.record(side)
    e = a + 5  # overlapped work on the capturing stream
    torch.cuda.current_stream().wait_event(ev)
    f = b + e
g2.replay()
g.replay()
g2.replay()
torch.cuda.synchronize()
say("   replayed:", bool(torch.equal(f, a * 3 + a + 5)), bool(torch.equal(d, a * 2 + 1)))
if max_stage < 8:
    sys.exit(0)
say("stage 8: torch.distributed nccl group of one beside it, OwnedComm through parallel.py ...")
dist.init_process_group("nccl", rank=0, world_size=1)
from chronoedit_amd.parallel import OwnedComm  # noqa: E402

oc = OwnedComm()
say("   OwnedComm up")
w = oc.all_to_all(a.view(1, -1), b.view(1, -1), async_op=True)
w.wait()
torch.cuda.synchronize()
say("   async all_to_all:", bool(torch.equal(a, b)))
oc.close()
lib.ce_comm_destroy(h)
dist.destroy_process_group()
say("done")
