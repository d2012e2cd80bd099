"""Ulysses sequence parallelism (+ CFG parallelism) for the DiT forward: one exchange step per self-attention.

Reference design (DiffSynth/xfuser path, not the diffusers path we sit behind):
chronoedit_diffsynth/wan_video_new_chronoedit.py:330-355 (USP enablement), :1448-1453 (token chunk + zero pad),
:1495-1498 (final all_gather); truncation re-shard: chronoedit/_src/models/chronoedit_14b_edit_model.py:168-186.
Everything in a block is token-wise except self-attention, so tokens are sharded contiguously (N/W per rank, last shard
zero-padded), weights replicated, and around the attention kernel q/k/v go from [N/W tokens, all 40 heads] to
[all N tokens, 40/W heads] and the attention output comes back.  RMSNorm-across-heads and RoPE are applied BEFORE the
exchange (they need all heads of a token / the token's position, both local).

MI355X layout (no permute().contiguous() passes anywhere on the path):
  * send side: `ce_rope_scatter_bf16` writes q | k | v of the local rows straight into the all-to-all SEND layout
    [dst rank][local row][tensor][D/W] while it normalises / rotates them (the pass that touched q and k anyway);
  * receive side: the buffer [src rank][row][tensor][D/W] IS [global token][tensor][D/W] (token shards are contiguous and
    rank-ordered), so the attention kernel reads q / k / v as strided views of it;
  * the attention output [global token][D/W] is already the send buffer of the second exchange (chunk t = rows of rank t);
  * its receive buffer [src rank = head group][local row][D/W] is consumed in place by the out-projection GEMM as a
    K-segmented A operand (`ce_gemm_aseg_bf16`).
Several samples per forward (the guidance pair batched, B = 2): local rows are [sample][local token], so the receive buffers are
[src rank][sample][local token] - token g of sample b in row (g // n) * B * n + b * n + g % n.  The attention kernel and the V
transposer take that BLOCKED layout as it is (`ce_attention_vt_blocked_bf16`, `ce_v_transpose_blocked_bf16`: one scalar
multiply-high per key tile; n is rounded up to a multiple of 64 so that no key tile straddles two source blocks), and the
attention output in the same layout is again the send buffer of the return exchange: twice the rows per GEMM, half the number of
collectives per step, still no permute pass.
The k|v exchange is issued asynchronously as soon as the k|v projection is done and overlaps the q projection GEMM
(SURVEY.md section 5.8); `torch.distributed` owns the communicator (backend "nccl" == RCCL over xGMI: an all-to-all drives
all 7 links of a GPU at once, each peer gets 1/W of the payload).  With the gloo backend (CPU tests, or several ranks sharing
one GPU in the GPU tests) CUDA tensors are staged through host memory.

CFG parallelism (SURVEY.md section 8e "additional free 2x"): the conditional and unconditional forwards of a guidance step are
independent, so a world of W ranks can run them as two Ulysses groups of W/2 ranks side by side (`CFGParallel`), then exchange
the two noise predictions (3.7 MB at N = 28 800) - half the exchange volume per rank and twice the rows per GEMM compared
with W-way Ulysses over sequential passes.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class _Done:
    def wait(self):
        return True


class _StreamJoin:
    """work.wait() of an exchange issued on the communicator's side stream: orders the CURRENT stream behind it (an event wait - legal
    under hipGraph capture, where it becomes the join edge of the fork the exchange was issued on)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        return True


class OwnedComm:
    """A RCCL communicator owned by libchronoedit_hip (csrc/ce_comm.hip, `ce_comm_*`) next to torch.distributed's own: the exchanges are
    plain C-ABI calls on a stream - no `Work` objects, no process-group watchdog polling them - so a sharded denoising step can be captured
    into a hipGraph (with torch's "nccl" backend the second capture kills the process: profiles/r03_rccl_graph_probe.txt).  SURVEY.md
    section 8b: "one ncclComm_t per process passed in".  The unique id travels over the existing process group (any backend); RCCL itself
    is the librccl.so torch ships, dlopen()ed by the library (one copy per process)."""

    _cache = {}  # (id of the process group OBJECT, device index) -> OwnedComm: ONE communicator per group and device for the life of the group

    @classmethod
    def get(cls, group: Optional[dist.ProcessGroup] = None, device: Optional[torch.device] = None) -> "OwnedComm":
        """The communicator of (group, device), created on first use and REUSED afterwards: a serving process that re-enables or re-groups
        sequence parallelism must not leave one ncclComm_t (device buffers, proxy threads) behind per call - ncclCommDestroy cannot be
        used on this stack (see close()).

        COLLECTIVE over `group` (every rank of the group calls it, as every rank constructs its Ulysses): the ranks first agree - one
        all_reduce(MIN) of a "my cached handle is usable" flag - whether ALL of them hold a communicator made for THIS group object with
        this world size and rank.  Only then is the cached one reused; otherwise every rank builds a new one together (a rank that called
        close(), or a process group that was destroyed and re-initialised with another world or rank, can therefore neither reuse a stale
        ncclComm_t nor enter ncclCommInitRank alone)."""
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        grp = group if group is not None else dist.group.WORLD  # the default group as an OBJECT: a re-initialised one is a different key
        key = (id(grp), dev.index)
        comm = cls._cache.get(key)
        usable = (comm is not None and comm.handle is not None and comm._group_ref is grp
                  and comm.world == dist.get_world_size(group) and comm.rank == dist.get_rank(group))
        flag = torch.tensor([1 if usable else 0], dtype=torch.int32, device=dev if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            if comm is not None:
                comm.handle = None  # (never destroyed: see close())
            for k in [k for k, c in cls._cache.items() if c._group_ref is not grp and k[1] == dev.index and c.handle is None]:
                del cls._cache[k]   # closed communicators of groups that are gone
            comm = cls._cache[key] = cls(group, dev)
        return comm

    def __init__(self, group: Optional[dist.ProcessGroup] = None, device: Optional[torch.device] = None):
        import ctypes
        import os

        from . import hiplib
        self.lib = hiplib.load()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            path = "librccl.so"  # the system copy (dlopen search path)
        if self.lib.ce_comm_load(path.encode()) != 0:
            raise RuntimeError(f"ce_comm_load({path}) failed")
        ident = [None]
        if self.rank == 0:
            buf = (ctypes.c_ubyte * 128)()
            if self.lib.ce_comm_unique_id(buf) != 0:
                raise RuntimeError("ce_comm_unique_id failed")
            ident = [bytes(buf)]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(ident, src=src, group=group)
        handle = ctypes.c_void_p()
        idb = (ctypes.c_ubyte * 128).from_buffer_copy(ident[0])
        with torch.cuda.device(self.device):
            if self.lib.ce_comm_init(ctypes.byref(handle), idb, self.rank, self.world) != 0:
                raise RuntimeError("ce_comm_init (ncclCommInitRank) failed")
        self.handle = handle
        self._group_ref = group if group is not None else dist.group.WORLD  # (also keeps the id() in the cache key unique while the entry lives)
        self.side = torch.cuda.Stream(device=self.device)  # the stream asynchronous exchanges run on (created outside any capture)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with code {rc} (RCCL's message is on stderr)")

    def all_to_all(self, send: torch.Tensor, recv: torch.Tensor, async_op: bool):
        nbytes = send.numel() * send.element_size() // self.world
        # Under hipGraph capture every exchange goes on the CAPTURING stream: a grouped ncclSend / ncclRecv issued on a stream that joined
        # the capture through an event (fork / join) takes the process down inside RCCL 2.26 (tools/owned_comm_probe.py, stage 7), while
        # any number of captures and replays with the exchanges on the capturing stream itself work.  The k|v-exchange / q-projection
        # overlap is an eager-mode nicety (it hides ~5 % of a layer at 8 GPUs: DESIGN.md section 6).
        if not async_op or torch.cuda.is_current_stream_capturing():
            self._check(self.lib.ce_comm_all_to_all(self.handle, send.data_ptr(), recv.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream),
                        "ce_comm_all_to_all")
            return _Done()
        self.side.wait_stream(torch.cuda.current_stream())  # fork: the exchange starts when the producer kernels of `send` are done
        self._check(self.lib.ce_comm_all_to_all(self.handle, send.data_ptr(), recv.data_ptr(), nbytes, self.side.cuda_stream), "ce_comm_all_to_all")
        ev = torch.cuda.Event()
        ev.record(self.side)
        return _StreamJoin(ev)

    def all_gather(self, x_local: torch.Tensor, out: torch.Tensor):
        self._check(self.lib.ce_comm_all_gather(self.handle, x_local.data_ptr(), out.data_ptr(), x_local.numel() * x_local.element_size(),
                                                torch.cuda.current_stream().cuda_stream), "ce_comm_all_gather")

    def close(self):
        """Forget the communicator (the cache then builds a new one on the next get()).  ncclCommDestroy is NOT called: on this stack (RCCL 2.26.6 beside torch's own process group) it blocks
        for good on a one-rank communicator (tools/owned_comm_probe.py); the handle lives until the process exits, like torch's own."""
        self.handle = None


class Ulysses:
    @property
    def sharded(self) -> bool:
        return self.world > 1 or self.force

    def __init__(self, group: Optional[dist.ProcessGroup] = None, force: bool = False, owned_comm: bool = False):
        """force: take the sharded code path even in a group of ONE rank (every exchange then runs as a real collective of the
        backend on a single rank) - how the RCCL call sequence is exercised on a one-GPU box (tests/test_ulysses.py).
        owned_comm: run the exchanges on a RCCL communicator owned by libchronoedit_hip (`OwnedComm`) instead of torch.distributed's -
        what makes the sharded step capturable into a hipGraph (`capturable`)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before enabling Ulysses sequence parallelism")
        self.group = group
        self.force = bool(force)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self._host_staged = self.backend == "gloo"
        self.stats = {"all_to_all_calls": 0, "all_to_all_bytes_sent_off_rank": 0, "all_gather_calls": 0}
        if owned_comm and self._host_staged:
            # ranks of a gloo group may share one GPU (the one-box test backend): ncclCommInitRank on duplicate devices fails or hangs
            raise RuntimeError("owned_comm=True needs the RCCL ('nccl') backend: under gloo the exchanges are host-staged and ranks may share a device")
        self.comm = OwnedComm.get(group) if owned_comm else None

    @property
    def capturable(self) -> bool:
        """Can a step with this group's exchanges be captured into a hipGraph?  Only on the library-owned communicator."""
        return self.comm is not None

    # -- token sharding ----------------------------------------------------------------
    def shard(self, n_tokens: int, align: int = 1) -> Tuple[int, int, int]:
        """(n_local, start, n_valid): every rank holds n_local = ceil(N / W) rows (rounded up to a multiple of `align`), rows past
        N are zero padding.  align = 64 is what the batched form needs: with several samples per forward the all-to-all receive
        buffer is [source rank][sample][local token], and a 64-key attention tile must not straddle two source blocks."""
        n_local = (n_tokens + self.world - 1) // self.world
        n_local = (n_local + align - 1) // align * align
        start = self.rank * n_local
        n_valid = max(0, min(n_local, n_tokens - start))
        return n_local, start, n_valid

    def take_rows(self, full: torch.Tensor, n_tokens: int, align: int = 1) -> torch.Tensor:
        """Local zero-padded slice [n_local, ...] of a replicated [N, ...] tensor."""
        n_local, start, n_valid = self.shard(n_tokens, align)
        out = torch.zeros((n_local,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
        if n_valid:
            out[:n_valid] = full[start : start + n_valid]
        return out

    # -- collectives -------------------------------------------------------------------
    def all_to_all(self, send: torch.Tensor, recv: Optional[torch.Tensor] = None, async_op: bool = False):
        """send / recv: contiguous, leading axis = peer rank.  Returns (recv, work); work.wait() orders the current stream
        behind the exchange (RCCL runs it on the process group's own stream, so kernels launched between the call and
        the wait overlap it)."""
        assert send.is_contiguous() and send.shape[0] == self.world
        if recv is None:
            recv = torch.empty_like(send)
        assert recv.is_contiguous() and recv.shape == send.shape
        self.stats["all_to_all_calls"] += 1
        self.stats["all_to_all_bytes_sent_off_rank"] += send.numel() * send.element_size() * (self.world - 1) // self.world
        if self.world == 1 and not self.force:
            recv.copy_(send)
            return recv, _Done()
        if self.comm is not None and send.is_cuda:
            return recv, self.comm.all_to_all(send, recv, async_op)
        if self._host_staged and send.is_cuda:
            sc = send.cpu()
            rc = torch.empty_like(sc)
            dist.all_to_all_single(rc, sc, group=self.group)
            recv.copy_(rc)
            return recv, _Done()
        work = dist.all_to_all_single(recv, send, group=self.group, async_op=async_op)
        return recv, (work if async_op else _Done())

    def all_gather_rows(self, x_local: torch.Tensor) -> torch.Tensor:
        """[n_local, C] -> [W*n_local, C] in rank order."""
        self.stats["all_gather_calls"] += 1
        if self.world == 1 and not self.force:
            return x_local
        x_local = x_local.contiguous()
        if self.comm is not None and x_local.is_cuda:
            out = torch.empty((self.world * x_local.shape[0],) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
            self.comm.all_gather(x_local, out)
            return out
        if self._host_staged and x_local.is_cuda:
            xc = x_local.cpu()
            out = torch.empty((self.world * xc.shape[0],) + tuple(xc.shape[1:]), dtype=xc.dtype)
            dist.all_gather_into_tensor(out, xc, group=self.group)
            return out.to(x_local.device)
        out = torch.empty((self.world * x_local.shape[0],) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
        dist.all_gather_into_tensor(out, x_local, group=self.group)
        return out

    # -- layout contracts (plain-torch statements of what the HIP kernels produce / consume; used by the CPU tests and as
    #    the fp8-mode fallback, never on the bf16 hot path) --------------------------------------------------------------
    def send_layout_reference(self, blocks: List[torch.Tensor]) -> torch.Tensor:
        """What ce_rope_scatter_bf16 writes for already normalised / rotated column blocks [n_local, D] each:
        [W(dst), n_local, len(blocks), D/W]."""
        W = self.world
        n_local, D = blocks[0].shape
        x = torch.stack([b.reshape(n_local, W, D // W) for b in blocks], dim=2)  # [n_local, W, nt, Dl]
        return x.permute(1, 0, 2, 3).contiguous()

    @staticmethod
    def gathered_view(recv: torch.Tensor) -> torch.Tensor:
        """[W(src), n_local, nt, Dl] -> [W*n_local (global token), nt*Dl] without a copy."""
        W, n_local, nt, Dl = recv.shape
        return recv.view(W * n_local, nt * Dl)

    @staticmethod
    def merge_heads_reference(y: torch.Tensor) -> torch.Tensor:
        """What the K-segmented GEMM operand means: [W(head group), n_local, Dl] -> [n_local, W*Dl]."""
        W, n_local, Dl = y.shape
        return y.permute(1, 0, 2).reshape(n_local, W * Dl)


class CFGParallel:
    """World of W = 2 * S ranks: ranks [0, S) run the conditional forward, ranks [S, W) the unconditional one, each as an
    S-way Ulysses group; `exchange` hands every rank both predictions.  Every rank must construct this (new_group is collective)."""

    def __init__(self):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before enabling CFG parallelism")
        W, r = dist.get_world_size(), dist.get_rank()
        if W % 2:
            raise ValueError(f"CFG parallelism needs an even world size, got {W}")
        S = W // 2
        self.world, self.rank, self.branch = W, r, r // S  # branch 0 = conditional, 1 = unconditional
        groups = [dist.new_group(list(range(b * S, (b + 1) * S))) for b in range(2)]
        self.sp_group = groups[self.branch]
        self.pairs = [dist.new_group([i, i + S]) for i in range(S)]
        self.pair_group = self.pairs[r % S]
        self._host_staged = dist.get_backend() == "gloo"

    def exchange(self, pred: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """pred = this rank's branch prediction (replicated inside its Ulysses group) -> (cond, uncond) on every rank."""
        x = pred.contiguous()
        n = x.shape[0]
        shape = (2 * n,) + tuple(x.shape[1:])  # concatenated along dim 0 (the form every backend accepts)
        if self._host_staged and x.is_cuda:
            xc = x.cpu()
            out = torch.empty(shape, dtype=xc.dtype)
            dist.all_gather_into_tensor(out, xc, group=self.pair_group)
            out = out.to(x.device)
        else:
            out = torch.empty(shape, dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(out, x, group=self.pair_group)
        return out[:n], out[n:]
