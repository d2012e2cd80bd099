"""Ulysses sequence parallelism for the DiT forward: one exchange step per self-attention.

Reference design (DiffSynth/xfuser path, not the diffusers path we sit behind):
chronoedit_diffsynth/wan_video_new_chronoedit.py:330-355 (USP enablement), :1448-1453 (token chunk + zero pad),
:1495-1498 (final all_gather).  Everything in a block is token-wise except self-attention, so tokens are sharded
contiguously (N/W per rank, last shard zero-padded), weights replicated, and around the attention kernel q/k/v go
from [N/W tokens, all 40 heads] to [all N tokens, 40/W heads] with ONE `all_to_all_single` (q, k, v fused in one
message) and the attention output comes back with a second one.  RMSNorm-across-heads and RoPE are applied BEFORE the
exchange (they need all heads of a token / the token's position, both local).

MI355X: xGMI is point-to-point, 7 links per GPU; an all-to-all drives all links at once (each peer gets 1/W of the
payload) — per layer and GPU at W = 8, N = 28 800: 3 x 36.9 MB out + 36.9 MB back.  `torch.distributed` owns the
communicator (backend "nccl" == RCCL); with the gloo backend (CPU tests, or two ranks sharing one GPU) tensors are
staged through host memory.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class Ulysses:
    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before enabling Ulysses sequence parallelism")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._host_staged = dist.get_backend(group) == "gloo"

    # -- token sharding ----------------------------------------------------------------
    def shard(self, n_tokens: int):
        """(n_local, start, n_valid): every rank holds n_local = ceil(N / W) rows, rows past N are zero padding."""
        n_local = (n_tokens + self.world - 1) // self.world
        start = self.rank * n_local
        n_valid = max(0, min(n_local, n_tokens - start))
        return n_local, start, n_valid

    def take_rows(self, full: torch.Tensor, n_tokens: int) -> torch.Tensor:
        """Local zero-padded slice [n_local, ...] of a replicated [N, ...] tensor."""
        n_local, start, n_valid = self.shard(n_tokens)
        out = torch.zeros((n_local,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
        if n_valid:
            out[:n_valid] = full[start : start + n_valid]
        return out

    # -- collectives -------------------------------------------------------------------
    def _a2a(self, x: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return x
        if self._host_staged and x.is_cuda:
            xc = x.cpu()
            yc = torch.empty_like(xc)
            dist.all_to_all_single(yc, xc, group=self.group)
            return yc.to(x.device)
        y = torch.empty_like(x)
        dist.all_to_all_single(y, x, group=self.group)
        return y

    def scatter_heads(self, qkv_local: torch.Tensor, heads: int, head_dim: int) -> torch.Tensor:
        """[n_local, 3*H*hd] (q|k|v, all heads) -> [W*n_local, 3*(H/W)*hd] (q|k|v of this rank's heads, all tokens)."""
        W = self.world
        n_local = qkv_local.shape[0]
        assert heads % W == 0, f"{heads} heads do not divide over {W} ranks"
        hl = heads // W
        x = qkv_local.view(n_local, 3, W, hl * head_dim).permute(2, 0, 1, 3).contiguous()  # [W(dst), n_local, 3, hl*hd]
        y = self._a2a(x)                                                                   # [W(src), n_local, 3, hl*hd]
        return y.view(W * n_local, 3 * hl * head_dim)

    def gather_heads(self, out_g: torch.Tensor, heads: int, head_dim: int) -> torch.Tensor:
        """[W*n_local, (H/W)*hd] (this rank's heads, all tokens) -> [n_local, H*hd] (all heads, local tokens)."""
        W = self.world
        hl = heads // W
        n_local = out_g.shape[0] // W
        x = out_g.view(W, n_local, hl * head_dim).contiguous()  # [W(dst token shard), n_local, hl*hd]
        y = self._a2a(x)                                        # [W(src head group), n_local, hl*hd]
        return y.permute(1, 0, 2).reshape(n_local, heads * head_dim).contiguous()

    def all_gather_rows(self, x_local: torch.Tensor) -> torch.Tensor:
        """[n_local, C] -> [W*n_local, C] in rank order."""
        if self.world == 1:
            return x_local
        x_local = x_local.contiguous()
        if self._host_staged and x_local.is_cuda:
            xc = x_local.cpu()
            out = torch.empty((self.world * xc.shape[0],) + tuple(xc.shape[1:]), dtype=xc.dtype)
            dist.all_gather_into_tensor(out, xc, group=self.group)
            return out.to(x_local.device)
        out = torch.empty((self.world * x_local.shape[0],) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
        dist.all_gather_into_tensor(out, x_local, group=self.group)
        return out
