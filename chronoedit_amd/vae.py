"""Wan-2.1 causal-3D-conv VAE on the gfx950 kernels — the `pipe.vae` slot of ChronoEditPipeline.

Interface parity (SURVEY.md §8b "VAE slot"; pipeline_chronoedit.py:119-129,185-186,427-434,442,672,765,776-781):
`.encode(x[B,3,F,H,W]).latent_dist.mode()`, `.decode(z, return_dict=False)[0]`,
`.config.{z_dim, latents_mean, latents_std}`, `.temperal_downsample`, `.dtype`.
Arithmetic spec: chronoedit/_src/tokenizers/wan2pt1.py:38-581 (the in-repo rendering of diffusers AutoencoderKLWan);
parameter names are that file's (`encoder.downsamples.3.residual.2.weight`, ...).

MI355X-first layout: activations are channels-last frames with a one-pixel zero border ([T][H+2][W+2][C] bf16), so
every conv of the network is the same implicit-GEMM kernel (`ce_conv_igemm_bf16`) with the taps as address offsets;
the causal temporal padding and the reference's chunk-to-chunk `feat_cache` (:571-580) are lists of frame POINTERS
(cache frames / a shared zero frame in front of the chunk), never concatenations.  The residual add of ResidualBlock
is fused into the second conv's epilogue, RMS_norm+SiLU is one pass, the temporal-upsample channel->frame interleave
(:137-139) is done by writing the two weight halves to the even / odd output frames.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Optional

import torch

from . import ops

CACHE_T = 2
LATENTS_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632,
                -0.1922, -0.9497, 0.2503, -0.2921]
LATENTS_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382,
               1.1253, 2.8251, 1.9160]


def _pad32(c):
    return (c + 31) // 32 * 32


class Frames:
    """T channels-last frames with a zero border: data [T, H+2, W+2, C] bf16.  front >= 0 / slack: the frames live inside a larger
    contiguous `stack` [front + T + 1] with `front` frames of room before them (the causal padding / cache frames of the conv that
    reads them) and one zeroed slack frame behind - the operand layout of ce_conv3d_gemm_bf16."""

    def __init__(self, T, H, W, C, device, data=None, front=None, zero=True):
        """zero=False: for a producer that writes every interior pixel and zeroes the border itself (ops.zero_border) - a full-resolution
        chunk is 0.7 GB, and zero-filling every buffer was 9 % of the decode.  The slack frame is zeroed here; the front frames are the
        consumer's to fill (cache frames or zeros)."""
        self.T, self.H, self.W, self.C = T, H, W, C
        self.stack, self.front = None, 0
        self.normed = None  # (gamma name, Frames): this activation after the NEXT layer's RMS_norm + SiLU, written by the conv that produced it
        alloc = torch.zeros if zero else torch.empty
        if front is not None:
            self.stack = alloc((front + T + 1, H + 2, W + 2, C), dtype=torch.bfloat16, device=device)
            if not zero:
                self.stack[front + T].zero_()
            self.front = front
            data = self.stack[front : front + T]
        self.data = data if data is not None else alloc((T, H + 2, W + 2, C), dtype=torch.bfloat16, device=device)

    def frame_list(self):
        return [self.data[t] for t in range(self.T)]

    def last(self, k):
        """The last k frames as a VIEW: a frame cache (wan2pt1.py:200-210) keeps its chunk's buffer alive instead of copying two frames of
        it - nothing writes into a buffer after the layer that produced it."""
        return self.data[self.T - k :]


class _ConvPack:
    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor], cin_pad=None, split_out=False):
        """[Cout, Cin, *k] (conv3d or conv2d) -> [Cout_pad8][taps][Cin_pad32] bf16 (+ fp32 bias)."""
        if w.dim() == 4:
            w = w.unsqueeze(2)
        Cout, Cin, KT, KH, KW = w.shape
        cin_p = cin_pad or _pad32(Cin)
        cout_p = (Cout + 7) // 8 * 8
        wp = torch.zeros((cout_p, KT * KH * KW, cin_p), dtype=torch.bfloat16, device=w.device)
        wp[:Cout, :, :Cin] = w.permute(0, 2, 3, 4, 1).reshape(Cout, KT * KH * KW, Cin).to(torch.bfloat16)
        self.w = wp.contiguous()
        self.b = None
        if b is not None:
            self.b = torch.zeros(cout_p, dtype=torch.float32, device=w.device)
            self.b[:Cout] = b.float()
        self.Cout, self.Cout_p, self.Cin_p, self.k = Cout, cout_p, cin_p, (KT, KH, KW)
        self._w_gemm = None

    def gemm_weight(self) -> torch.Tensor:
        """[Cout_pad8][K-tiles rounded up to even x 64]: the same taps-major rows, zero columns behind them (ce_conv3d_gemm_bf16)."""
        if self._w_gemm is None:
            KT, KH, KW = self.k
            cout, cin = self.w.shape[0], self.w.shape[2]
            seg = (KW * cin + 63) // 64 * 64  # one (kt, kh) run: three pixels' channels, whole K-tiles (Cin = 96: 288 -> 320)
            k = KT * KH * seg
            kpad = ((k // 64) + 1) // 2 * 2 * 64
            wg = torch.zeros((cout, kpad), dtype=torch.bfloat16, device=self.w.device)
            wg[:, :k].view(cout, KT * KH, seg)[:, :, : KW * cin] = self.w.reshape(cout, KT * KH, KW * cin)
            self._w_gemm = wg
        return self._w_gemm


class WanVAEEngine:
    def __init__(self, params: Dict[str, torch.Tensor], dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2,
                 temperal_downsample=(False, True, True), temporal_window=4):
        self.cfg = SimpleNamespace(dim=dim, z_dim=z_dim, dim_mult=tuple(dim_mult), num_res_blocks=num_res_blocks,
                                   temperal_downsample=tuple(temperal_downsample), temporal_window=temporal_window)
        some = next(iter(params.values()))
        if not some.is_cuda:
            raise ops.HipKernelError("WanVAEEngine needs its parameters on the GPU (no CPU fallback)")
        self.dev = some.device
        self.p = params
        self.packs: Dict[str, _ConvPack] = {}
        self.gammas: Dict[str, torch.Tensor] = {}
        for k, v in params.items():
            if k.endswith(".weight"):
                name = k[: -len(".weight")]
                self.packs[name] = _ConvPack(v, params.get(name + ".bias"))
            elif k.endswith("gamma"):
                self.gammas[k] = v.float().reshape(-1).contiguous()
        self._zero: Dict[tuple, torch.Tensor] = {}
        self._attn_vt: Dict[tuple, torch.Tensor] = {}  # mid-block attention scratch per (C, padded h*w): see _attn
        self._attn_ws: Dict[tuple, tuple] = {}
        self.use_gemm_conv = True  # wide stride-1 3x3(x3) convs on the large-tile GEMM (False: every conv on the implicit-GEMM kernel)
        self.use_head_conv = True  # the decoder's 96 -> 3 head conv on its own bandwidth kernel (False: implicit GEMM with N = 8)
        self.fuse_norm = True      # 96-channel ResidualBlocks: the second RMS_norm + SiLU in the first conv's epilogue (False: its own pass)
        self._layers()

    # -- architecture (wan2pt1.py:283-305, 384-415) ------------------------------------------------
    def _layers(self):
        c = self.cfg
        dims = [c.dim * u for u in (1,) + c.dim_mult]
        enc, idx = [], 0
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(c.num_res_blocks):
                enc.append(("res", f"encoder.downsamples.{idx}", cin, cout))
                idx += 1
                cin = cout
            if i != len(c.dim_mult) - 1:
                enc.append(("down3d" if c.temperal_downsample[i] else "down2d", f"encoder.downsamples.{idx}", cout))
                idx += 1
        ce = dims[-1]
        self.enc = enc + [("res", "encoder.middle.0", ce, ce), ("attn", "encoder.middle.1", ce), ("res", "encoder.middle.2", ce, ce)]
        up = c.temperal_downsample[::-1]
        ddims = [c.dim * u for u in (c.dim_mult[-1],) + c.dim_mult[::-1]]
        c0 = ddims[0]
        dec = [("res", "decoder.middle.0", c0, c0), ("attn", "decoder.middle.1", c0), ("res", "decoder.middle.2", c0, c0)]
        idx = 0
        for i, (cin, cout) in enumerate(zip(ddims[:-1], ddims[1:])):
            if i in (1, 2, 3):
                cin //= 2
            for _ in range(c.num_res_blocks + 1):
                dec.append(("res", f"decoder.upsamples.{idx}", cin, cout))
                idx += 1
                cin = cout
            if i != len(c.dim_mult) - 1:
                dec.append(("up3d" if up[i] else "up2d", f"decoder.upsamples.{idx}", cout))
                idx += 1
        self.dec = dec

    # -- primitives ----------------------------------------------------------------------------------
    def _zero_frame(self, H, W, C):
        key = (H, W, C)
        if key not in self._zero:
            self._zero[key] = torch.zeros((H + 2, W + 2, C), dtype=torch.bfloat16, device=self.dev)
        return self._zero[key]

    def _conv(self, name, in_frames: List[torch.Tensor], T_out, H_out, W_out, in_W, *, st=1, ss=1, in_off=0, res: Optional[Frames] = None,
              out: Optional[Frames] = None, out_frames=None, rows=None, cout_slice=None, out_C=None):
        """Run conv `name` over a list of input frames; returns the output Frames (bordered) unless `rows` is given
        (an un-bordered [T*H*W, C] matrix written in place)."""
        pk = self.packs[name]
        KT, KH, KW = pk.k
        w, b, Cout = pk.w, pk.b, pk.Cout_p
        if cout_slice is not None:  # a contiguous range of output channels (temporal-upsample halves)
            lo, hi = cout_slice
            w, b, Cout = pk.w[lo:hi], (None if pk.b is None else pk.b[lo:hi].contiguous()), hi - lo
        if rows is not None:
            ops.conv_igemm(in_frames, w, b, [rows[t] for t in range(T_out)], None, Cin=pk.Cin_p, Cout=Cout, KT=KT, KH=KH, KW=KW, st=st,
                           ss=ss, H_out=H_out, W_out=W_out, in_Wp=in_W + 2, in_off=in_off, out_Wp=W_out, out_border=0,
                           out_cstride=rows.shape[-1])
            return rows
        fresh = out is None and out_frames is None
        if fresh:  # every interior pixel of channels [0, Cout) is written: zero-fill only when the buffer is wider than that
            out = Frames(T_out, H_out, W_out, out_C or Cout, self.dev, zero=(out_C or Cout) != Cout)
        of = out_frames if out_frames is not None else out.frame_list()
        oC = out.C if out is not None else out_C
        if (self.use_head_conv and pk.Cout <= 4 and pk.Cin_p == 96 and (KH, KW) == (3, 3) and KT in (1, 3) and st == 1 and ss == 1
                and in_off == 0 and res is None and cout_slice is None):
            # the decoder's 96 -> 3 head conv: three channels do not fill an implicit-GEMM tile (ce_conv3d_head_bf16)
            ops.conv3d_head(in_frames, w, b, of, Cin=96, Cout=pk.Cout, KT=KT, H_out=H_out, W_out=W_out, in_Wp=in_W + 2, out_Wp=W_out + 2,
                            out_border=1, out_cstride=oC)
        else:
            ops.conv_igemm(in_frames, w, b, of, res.frame_list() if res is not None else None, Cin=pk.Cin_p, Cout=Cout, KT=KT, KH=KH, KW=KW,
                           st=st, ss=ss, H_out=H_out, W_out=W_out, in_Wp=in_W + 2, in_off=in_off, out_Wp=W_out + 2, out_border=1,
                           out_cstride=oC)
        if fresh and out.C == Cout:
            ops.zero_border(out.data, T_out, H_out, W_out, Cout)
        return out

    def _gemm_ok(self, name, C_in, st: int = 1, ss: int = 1) -> bool:
        """Does conv `name` run on the large-tile GEMM (ce_conv3d_gemm_bf16)?  Stride-1 3x3(x3), at least 96 channels in and out.
        (The strided resample convs share the kernel size: the stride is part of the question.)"""
        pk = self.packs[name]
        return (self.use_gemm_conv and st == 1 and ss == 1 and pk.k in ((3, 3, 3), (1, 3, 3)) and pk.Cin_p == C_in and pk.Cout_p >= 96
                and (C_in >= 96 or (C_in == 32 and pk.Cout_p == 96)))  # (32 -> 96: the encoder's stem, on the slab kernel of the 96-channel layers)

    def _conv_gemm(self, name, x: Frames, front_frames, res: Optional[Frames], out_C=None, norm=None, also_norm=None) -> Frames:
        """norm = (gamma name, front): the NEXT layer's RMS_norm + SiLU in this conv's epilogue (ce_conv3d_gemm_rms_silu_bf16; the output is
        what _rms_silu(conv output, gamma, front=front) would have produced, and the un-normalised activation is never written).
        also_norm = (gamma name, front): the conv's result as usual, plus that normalised form as `result.normed` (taken when the kernel can)."""
        pk = self.packs[name]
        KT = pk.k[0]
        need = KT - 1
        if x.stack is None or x.front != need:  # the producer did not leave room: one copy of the chunk into a stack
            st = Frames(x.T, x.H, x.W, x.C, self.dev, front=need)
            st.data.copy_(x.data)
            x = st
        zero = self._zero.get((x.H, x.W, x.C))
        for j, f in enumerate(front_frames):
            if f is zero:
                x.stack[j].zero_()
            else:
                x.stack[j].copy_(f)
        if norm is not None:
            assert res is None and out_C is None and self._norm_fusable(name)
            out = Frames(x.T, x.H, x.W, pk.Cout_p, self.dev, front=norm[1], zero=False)
            ops.conv3d_gemm_rms_silu(x.stack, pk.gemm_weight(), pk.b, out.data, self.gammas[norm[0]], T_out=x.T, H=x.H, W=x.W, Cin=pk.Cin_p,
                                     Cout=pk.Cout_p, KT=KT)
            return out
        if also_norm is not None and out_C is None and self._norm_fusable(name):
            # the result AND its normalised form (the next layer's first norm): two outputs of one launch, the re-reading pass is gone
            out = Frames(x.T, x.H, x.W, pk.Cout_p, self.dev, zero=False)
            nrm = Frames(x.T, x.H, x.W, pk.Cout_p, self.dev, front=also_norm[1], zero=False)
            ops.conv3d_gemm_rms_silu(x.stack, pk.gemm_weight(), pk.b, nrm.data, self.gammas[also_norm[0]], T_out=x.T, H=x.H, W=x.W, Cin=pk.Cin_p,
                                     Cout=pk.Cout_p, KT=KT, out_stack=out.data, res_stack=res.data if res is not None else None)
            out.normed = (also_norm[0], nrm)
            return out
        out = Frames(x.T, x.H, x.W, out_C or pk.Cout_p, self.dev, zero=(out_C or pk.Cout_p) != pk.Cout_p)  # (the kernel zeroes the border)
        ops.conv3d_gemm(x.stack, pk.gemm_weight(), pk.b, out.data, res.data if res is not None else None, T_out=x.T, H=x.H, W=x.W,
                        Cin=pk.Cin_p, Cout=pk.Cout_p, KT=KT)
        return out

    def _norm_fusable(self, name) -> bool:
        """Can conv `name` take the next RMS_norm + SiLU into its epilogue?  The slab kernel of the 96-channel layers only."""
        pk = self.packs[name]
        return self.fuse_norm and pk.Cout_p == 96 and pk.Cin_p in (32, 96, 192) and pk.k in ((3, 3, 3), (1, 3, 3))

    def _cached_conv(self, name, x: Frames, caches, res=None, out_C=None, gemm: Optional[bool] = None, norm=None, also_norm=None) -> Frames:
        """3x3x3 causal conv with the chunk-to-chunk frame cache (wan2pt1.py:200-210): two frames in front of the chunk.
        gemm: the routing decision when the caller already took it (the producer sized x's stack by it); None: decide here."""
        i = caches["i"]
        caches["i"] += 1
        prev = caches["slots"].get(i)
        z = self._zero_frame(x.H, x.W, x.C)
        if prev is None:
            front = [z, z]
        elif prev.shape[0] == 1:
            front = [z, prev[0]]
        else:
            front = [prev[0], prev[1]]
        if x.T >= CACHE_T:
            keep = x.last(CACHE_T)
        elif prev is not None:
            keep = torch.cat([prev[-1:], x.data], 0)
        else:
            keep = x.data.clone()
        if gemm is None:
            gemm = self._gemm_ok(name, x.C)
        if gemm:
            out = self._conv_gemm(name, x, front, res, out_C, norm=norm, also_norm=also_norm)
        else:
            assert norm is None
            out = self._conv(name, front + x.frame_list(), x.T, x.H, x.W, x.W, res=res, out_C=out_C)
        caches["slots"][i] = keep
        return out

    def _rms_silu(self, x: Frames, gname, silu=True, border=1, front=None) -> Frames:
        out = Frames(x.T, x.H, x.W, x.C, self.dev, front=front, zero=False) if border else None
        if border:
            ops.rms_silu(x.data, self.gammas[gname], out.data, x.T, x.C, x.H, x.W, 1, 1, silu)
            ops.zero_border(out.data, x.T, x.H, x.W, x.C)
            return out
        rows = torch.empty((x.T, x.H * x.W, x.C), dtype=torch.bfloat16, device=self.dev)
        ops.rms_silu(x.data, self.gammas[gname], rows, x.T, x.C, x.H, x.W, 1, 0, silu)
        return rows

    def _first_norm(self, x: Frames, gname, front) -> Frames:
        """RMS_norm + SiLU of x - already there when the conv that produced x wrote it as a second output (Frames.normed)."""
        if x.normed is not None and x.normed[0] == gname and (x.normed[1].front if x.normed[1].stack is not None else None) == front:
            return x.normed[1]
        return self._rms_silu(x, gname, front=front)

    def _res(self, name, x: Frames, cin, cout, caches, next_norm=None) -> Frames:
        """next_norm = (gamma name, front) of the layer that consumes this block's output through an RMS_norm + SiLU first (the next
        ResidualBlock or the head): handed to the block's last conv, which writes that normalised form beside its result when it can."""
        h = x
        if (name + ".shortcut") in self.packs:
            h = self._conv(name + ".shortcut", x.frame_list(), x.T, x.H, x.W, x.W, in_off=1)
        g2 = self._gemm_ok(name + ".residual.2", x.C)  # one routing decision per layer: it sizes the producer's stack AND picks the conv
        y = self._first_norm(x, name + ".residual.0.gamma", 2 if g2 else None)
        g6 = self._gemm_ok(name + ".residual.6", self.packs[name + ".residual.2"].Cout_p)
        if g2 and self._norm_fusable(name + ".residual.2"):
            # the first conv's output feeds nothing but the second norm (wan2pt1.py:195-200): norm + SiLU ride in the conv's epilogue
            y = self._cached_conv(name + ".residual.2", y, caches, gemm=True, norm=(name + ".residual.3.gamma", 2 if g6 else None))
        else:
            y = self._cached_conv(name + ".residual.2", y, caches, gemm=g2)
            y = self._rms_silu(y, name + ".residual.3.gamma", front=2 if g6 else None)
        return self._cached_conv(name + ".residual.6", y, caches, res=h, gemm=g6, also_norm=next_norm if (g6 and self.fuse_norm) else None)

    def _attn(self, name, x: Frames) -> Frames:
        """Per-frame single-head attention over h*w (wan2pt1.py:240-259): 1x1 qkv conv -> one flash-style attention kernel
        (ce_attention_1head_bf16) -> 1x1 proj conv with the identity fused as residual."""
        C, HW = x.C, x.H * x.W
        xn = self._rms_silu(x, name + ".norm.gamma", silu=False)  # bordered
        qkv = torch.empty((x.T, HW, 3 * C), dtype=torch.bfloat16, device=self.dev)
        self._conv(name + ".to_qkv", xn.frame_list(), x.T, x.H, x.W, x.W, in_off=1, rows=qkv)
        hwp = (HW + 63) // 64 * 64
        o = torch.empty((x.T, HW, C), dtype=torch.bfloat16, device=self.dev)
        if C in (128, 384):  # the shipped width (384) and the dim-32 test width: one flash-style kernel per frame, nothing [HW, HW]-sized
            # one V^T scratch PER SHAPE, never released while the engine lives: a hipGraph captured at one resolution keeps the raw
            # pointer of its scratch, so rebinding a single engine-wide buffer when another resolution comes along would hand the
            # captured graph freed memory (and its padding columns must stay zero)
            vt = self._attn_vt.get((C, hwp))
            if vt is None:
                vt = self._attn_vt[(C, hwp)] = torch.zeros((C, hwp), dtype=torch.bfloat16, device=self.dev)  # padding columns stay zero
            for t in range(x.T):
                vt[:, :HW].copy_(qkv[t, :, 2 * C :].t())
                ops.attention_1head(qkv[t, :, :C], qkv[t, :, C : 2 * C], vt, C ** -0.5, out=o[t])
        else:
            # other widths: query rows in chunks - the fp32 score block is [rows, HW] with rows chosen for <= 256 MiB, and the work
            # buffers live across frames and calls (scores, probabilities, V^T with its padding columns zeroed once)
            rows = min(HW, max(256, (1 << 26) // hwp // 64 * 64))
            ws = self._attn_ws.get((rows, hwp, C))  # keyed by shape and kept, like _attn_vt above
            if ws is None:
                ws = ((rows, hwp, C), torch.empty((rows, hwp), dtype=torch.float32, device=self.dev),
                      torch.empty((rows, hwp), dtype=torch.bfloat16, device=self.dev), torch.zeros((C, hwp), dtype=torch.bfloat16, device=self.dev))
                self._attn_ws[(rows, hwp, C)] = ws
            _, s_buf, p_buf, vt = ws
            for t in range(x.T):
                q, k, v = qkv[t, :, :C], qkv[t, :, C : 2 * C], qkv[t, :, 2 * C :]
                vt[:, :HW].copy_(v.t())
                for r0 in range(0, HW, rows):
                    n = min(rows, HW - r0)
                    ops.gemm_f32(q[r0:r0 + n], k, out=s_buf[:n, :HW])  # [n, HW] fp32
                    ops.softmax_rows(s_buf[:n, :HW], p_buf[:n], HW, C ** -0.5)
                    ops.gemm(p_buf[:n], vt, None, out=o[t, r0:r0 + n])
        # proj (1x1) on the un-bordered rows, + identity, into a bordered stack
        out = Frames(x.T, x.H, x.W, C, self.dev)
        pk = self.packs[name + ".proj"]
        ops.conv_igemm([o[t] for t in range(x.T)], pk.w, pk.b, out.frame_list(), x.frame_list(), Cin=pk.Cin_p, Cout=pk.Cout_p, KT=1, KH=1,
                       KW=1, st=1, ss=1, H_out=x.H, W_out=x.W, in_Wp=x.W, in_off=0, out_Wp=x.W + 2, out_border=1, out_cstride=C)
        return out

    def _down(self, kind, name, x: Frames, caches) -> Frames:
        """Resample downsample2d/3d (wan2pt1.py:108-112,150-165)."""
        y = self._conv(name + ".resample.1", x.frame_list(), x.T, x.H // 2, x.W // 2, x.W, ss=2, in_off=1)
        if kind == "down3d":
            i = caches["i"]
            caches["i"] += 1
            prev = caches["slots"].get(i)
            if prev is None:
                caches["slots"][i] = y.data.clone()
            else:
                frames = [prev[-1]] + y.frame_list()
                keep = y.last(1)
                T_out = (len(frames) - 3) // 2 + 1
                y2 = self._conv(name + ".time_conv", frames, T_out, y.H, y.W, y.W, st=2, in_off=1)
                caches["slots"][i] = keep
                y = y2
        return y

    def _up(self, kind, name, x: Frames, caches, next_norm=None) -> Frames:
        """Resample upsample2d/3d (wan2pt1.py:99-104,118-139).  next_norm: see _res (the upsample's conv is the producer here)."""
        C = x.C
        if kind == "up3d":
            i = caches["i"]
            caches["i"] += 1
            prev = caches["slots"].get(i)
            if prev is None:
                caches["slots"][i] = "Rep"
            else:
                z = self._zero_frame(x.H, x.W, C)
                if isinstance(prev, str):
                    front = [z, z]
                else:
                    front = [prev[0], prev[1]]
                if x.T >= CACHE_T:
                    keep = x.last(CACHE_T)
                elif isinstance(prev, str):
                    keep = torch.cat([z[None], x.data], 0)
                else:
                    keep = torch.cat([prev[-1:], x.data], 0)
                y = Frames(2 * x.T, x.H, x.W, C, self.dev)
                fl = y.frame_list()
                ins = front + x.frame_list()
                # time_conv has 2C output channels: channels [0,C) are frame 2t, [C,2C) frame 2t+1 (:137-139)
                self._conv(name + ".time_conv", ins, x.T, x.H, x.W, x.W, in_off=1, out_frames=fl[0::2], cout_slice=(0, C), out_C=C)
                self._conv(name + ".time_conv", ins, x.T, x.H, x.W, x.W, in_off=1, out_frames=fl[1::2], cout_slice=(C, 2 * C), out_C=C)
                caches["slots"][i] = keep
                x = y
        gemm = self._gemm_ok(name + ".resample.1", C)
        u = Frames(x.T, 2 * x.H, 2 * x.W, C, self.dev, front=0 if gemm else None, zero=False)
        ops.upsample2x(x.data, u.data, x.T, C, x.H, x.W)
        ops.zero_border(u.data, u.T, u.H, u.W, C)
        if gemm:
            return self._conv_gemm(name + ".resample.1", u, [], None, also_norm=next_norm if self.fuse_norm else None)
        return self._conv(name + ".resample.1", u.frame_list(), u.T, u.H, u.W, u.W)

    def _run(self, layers, x, caches, tail_norm=None):
        """tail_norm: the gamma of the RMS_norm + SiLU that follows the last layer (the head's)."""
        for i, l in enumerate(layers):
            kind = l[0]
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            nn_ = None
            if kind in ("res", "up2d", "up3d"):
                last = l[1] + (".residual.6" if kind == "res" else ".resample.1")  # the conv whose epilogue would carry the next norm
                if nxt is not None and nxt[0] == "res":  # the next block's first norm; its conv's routing decides the stack the norm leaves room in
                    nn_ = (nxt[1] + ".residual.0.gamma", 2 if self._gemm_ok(nxt[1] + ".residual.2", self.packs[last].Cout_p) else None)
                elif nxt is None and tail_norm is not None:
                    nn_ = tail_norm
            if kind == "res":
                x = self._res(l[1], x, l[2], l[3], caches, next_norm=nn_)
            elif kind == "attn":
                x = self._attn(l[1], x)
            elif kind in ("down2d", "down3d"):
                x = self._down(kind, l[1], x, caches)
            else:
                x = self._up(kind, l[1], x, caches, next_norm=nn_)
        return x

    def _to_frames(self, x: torch.Tensor, C_pad: int, front=None) -> Frames:
        """[C, T, H, W] (any float dtype) -> bordered channels-last bf16 frames with channels zero-padded to C_pad (front: room for the
        consumer's cache frames, see Frames)."""
        C, T, H, W = x.shape
        f = Frames(T, H, W, C_pad, self.dev, front=front)
        f.data[:, 1 : H + 1, 1 : W + 1, :C] = x.permute(1, 2, 3, 0).to(torch.bfloat16)
        return f

    # -- encode / decode (wan2pt1.py:502-560) -----------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """x [3, T, H, W] -> mu [z, T', H/8, W/8] (un-normalised mode of the posterior), chunks of 1, 4, 4, ... frames."""
        c = self.cfg
        T = x.shape[1]
        caches = {"slots": {}, "i": 0}
        chunks = [x[:, :1]]
        n = 1 + (T - 1) // c.temporal_window
        for i in range(1, n):
            chunks.append(x[:, 1 + c.temporal_window * (i - 1) : 1 + c.temporal_window * i])
        if (T - 1) % c.temporal_window:
            chunks.append(x[:, 1 + c.temporal_window * (n - 1) :])
        outs = []
        for ch in chunks:
            caches["i"] = 0
            g1 = self._gemm_ok("encoder.conv1", 32)
            f = self._to_frames(ch, 32, front=2 if g1 else None)
            first = self.enc[0]  # the stem feeds the first ResidualBlock's shortcut and, through its first norm, its first conv
            an = None
            if g1 and self.fuse_norm and first[0] == "res":
                an = (first[1] + ".residual.0.gamma", 2 if self._gemm_ok(first[1] + ".residual.2", self.packs["encoder.conv1"].Cout_p) else None)
            f = self._cached_conv("encoder.conv1", f, caches, gemm=g1, also_norm=an)
            gh = self._gemm_ok("encoder.head.2", self.packs["encoder.head.2"].Cin_p)
            f = self._run(self.enc, f, caches, tail_norm=("encoder.head.0.gamma", 2 if gh else None))
            f = self._first_norm(f, "encoder.head.0.gamma", 2 if gh else None)
            f = self._cached_conv("encoder.head.2", f, caches, gemm=gh)  # 2*z channels
            f = self._conv("conv1", f.frame_list(), f.T, f.H, f.W, f.W, in_off=1)
            outs.append(f.data[:, 1:-1, 1:-1, : c.z_dim])
        mu = torch.cat(outs, 0)  # [T', h, w, z]
        return mu.permute(3, 0, 1, 2).contiguous()

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [z_dim, T', h, w] (already de-normalised) -> video [3, T, 8h, 8w], one latent frame at a time."""
        c = self.cfg
        caches = {"slots": {}, "i": 0}
        zin = self._to_frames(z, 32)
        x = self._conv("conv2", zin.frame_list(), zin.T, zin.H, zin.W, zin.W, in_off=1, out_C=32)  # 16 live + 16 zero channels
        outs = []
        for i in range(x.T):
            caches["i"] = 0
            f = Frames(1, x.H, x.W, x.C, self.dev, data=x.data[i : i + 1])
            f = self._cached_conv("decoder.conv1", f, caches)
            f = self._run(self.dec, f, caches, tail_norm=("decoder.head.0.gamma", None))
            f = self._first_norm(f, "decoder.head.0.gamma", None)
            f = self._cached_conv("decoder.head.2", f, caches)  # 3 (+5 pad) channels
            outs.append(f.data[:, 1:-1, 1:-1, :3])
        v = torch.cat(outs, 0)  # [T, H, W, 3]
        return v.permute(3, 0, 1, 2).contiguous()


def wan_vae_param_shapes(dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_downsample=(False, True, True)):
    """{state-dict key: shape} of the Wan 2.1 VAE in the reference's own naming (_src/tokenizers/wan2pt1.py:283-305,384-415,
    492-493): what a checkpoint for `AutoencoderKLWan(params)` must contain."""
    eng = WanVAEEngine.__new__(WanVAEEngine)
    eng.cfg = SimpleNamespace(dim=dim, z_dim=z_dim, dim_mult=tuple(dim_mult), num_res_blocks=num_res_blocks,
                              temperal_downsample=tuple(temperal_downsample), temporal_window=4)
    eng._layers()
    s: Dict[str, tuple] = {}

    def conv(name, cin, cout, k):
        s[name + ".weight"] = (cout, cin) + tuple(k)
        s[name + ".bias"] = (cout,)

    def block(l):
        kind, name = l[0], l[1]
        if kind == "res":
            cin, cout = l[2], l[3]
            s[name + ".residual.0.gamma"] = (cin, 1, 1, 1)
            conv(name + ".residual.2", cin, cout, (3, 3, 3))
            s[name + ".residual.3.gamma"] = (cout, 1, 1, 1)
            conv(name + ".residual.6", cout, cout, (3, 3, 3))
            if cin != cout:
                conv(name + ".shortcut", cin, cout, (1, 1, 1))
        elif kind == "attn":
            c = l[2]
            s[name + ".norm.gamma"] = (c, 1, 1)
            conv(name + ".to_qkv", c, 3 * c, (1, 1))
            conv(name + ".proj", c, c, (1, 1))
        elif kind in ("down2d", "down3d"):
            c = l[2]
            conv(name + ".resample.1", c, c, (3, 3))
            if kind == "down3d":
                conv(name + ".time_conv", c, c, (3, 1, 1))
        else:  # up2d / up3d
            c = l[2]
            conv(name + ".resample.1", c, c // 2, (3, 3))
            if kind == "up3d":
                conv(name + ".time_conv", c, 2 * c, (3, 1, 1))

    conv("encoder.conv1", 3, dim, (3, 3, 3))
    for l in eng.enc:
        block(l)
    ec = dim * dim_mult[-1]
    s["encoder.head.0.gamma"] = (ec, 1, 1, 1)
    conv("encoder.head.2", ec, 2 * z_dim, (3, 3, 3))
    conv("conv1", 2 * z_dim, 2 * z_dim, (1, 1, 1))
    conv("conv2", z_dim, z_dim, (1, 1, 1))
    conv("decoder.conv1", z_dim, ec, (3, 3, 3))
    for l in eng.dec:
        block(l)
    s["decoder.head.0.gamma"] = (dim, 1, 1, 1)
    conv("decoder.head.2", dim, 3, (3, 3, 3))
    return s


class AutoencoderKLWan(torch.nn.Module):
    """`pipe.vae`-compatible wrapper around WanVAEEngine (parameters held as a flat buffer dict)."""

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, torch_dtype: torch.dtype = torch.bfloat16, device="cuda:0",
                        **unused):
        """``AutoencoderKLWan.from_pretrained(model_path, subfolder="vae", torch_dtype=bf16)`` (run_inference_diffusers.py:341-345):
        config.json (base_dim, z_dim, dim_mult, num_res_blocks, temperal_downsample) + safetensors in the diffusers or the native
        Wan naming; the key set and every shape are checked against the architecture."""
        from . import weights
        cfg = weights.read_config(path, subfolder)
        arch = dict(dim=cfg.get("base_dim", 96), z_dim=cfg.get("z_dim", 16), dim_mult=tuple(cfg.get("dim_mult", (1, 2, 4, 4))),
                    num_res_blocks=cfg.get("num_res_blocks", 2), temperal_downsample=tuple(cfg.get("temperal_downsample", (False, True, True))))
        sd = weights.load_state_dict_files(weights.shard_files(path, subfolder))
        sd = weights.wan_vae_diffusers_to_native(sd, arch["num_res_blocks"], len(arch["dim_mult"]))
        want = wan_vae_param_shapes(**arch)
        if set(sd) != set(want):
            odd = sorted(set(sd) ^ set(want))
            raise KeyError(f"VAE checkpoint does not match the architecture: {odd[:6]}")
        for k, shp in want.items():
            if tuple(sd[k].shape) != tuple(shp) and sd[k].numel() != int(torch.tensor(shp).prod()):
                raise ValueError(f"{k}: checkpoint shape {tuple(sd[k].shape)} != {tuple(shp)}")
        return cls({k: sd[k].reshape(want[k]).to(device) for k in want}, **arch)

    @classmethod
    def random_init(cls, device, seed: int = 0, **arch):
        """The architecture with seeded random weights (convs ~ N(0, 1/fan_in), gammas 1): for benchmarks without a checkpoint."""
        g = torch.Generator(device=device).manual_seed(seed)
        params = {}
        for k, shp in wan_vae_param_shapes(**arch).items():
            if k.endswith("gamma"):
                params[k] = torch.ones(shp, device=device)
            elif k.endswith(".bias"):
                params[k] = torch.zeros(shp, device=device)
            else:
                fan_in = 1
                for d in shp[1:]:
                    fan_in *= d
                params[k] = torch.randn(shp, generator=g, device=device) / fan_in ** 0.5
        return cls(params, **arch)

    def __init__(self, params: Dict[str, torch.Tensor], dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2,
                 temperal_downsample=(False, True, True)):
        super().__init__()
        self.config = SimpleNamespace(z_dim=z_dim, latents_mean=LATENTS_MEAN[:z_dim], latents_std=LATENTS_STD[:z_dim], base_dim=dim,
                                      dim_mult=list(dim_mult), num_res_blocks=num_res_blocks,
                                      temperal_downsample=list(temperal_downsample))
        self.temperal_downsample = list(temperal_downsample)
        self._params = params
        self._engine = None
        # use_graph: from the second call with a given input shape on, encode / decode replay ONE captured hipGraph (several hundred
        # launches and allocations per call otherwise: at 720p the decode is 49 ms of kernels in 70 ms of wall time).  The first call of
        # a shape runs eagerly - it also performs every lazy initialisation - so a one-off call pays nothing.
        self.use_graph = False
        self._graphs: Dict[tuple, object] = {}
        self._graph_pool = None

    @property
    def dtype(self):
        return torch.bfloat16

    def engine(self) -> WanVAEEngine:
        if self._engine is None:
            c = self.config
            self._engine = WanVAEEngine(self._params, c.base_dim, c.z_dim, tuple(c.dim_mult), c.num_res_blocks, tuple(c.temperal_downsample))
        return self._engine

    MAX_GRAPHS = 4  # captured shapes kept at once; the least recently used one is dropped for a new shape

    def clear_graphs(self):
        """Drop every captured encode / decode graph with its static buffers and activation pool (a full-resolution chunk is ~0.7 GB
        per buffer).  The next call of a shape runs eagerly again, the one after re-captures.  Called by the pipeline's
        `offload_model`; call it yourself when a serving process is done with a resolution."""
        self._graphs.clear()
        self._graph_pool = None

    def _run(self, kind: str, fn, x: torch.Tensor) -> torch.Tensor:
        if not self.use_graph or not x.is_cuda:
            return fn(x)
        key = (kind, tuple(x.shape), x.dtype)
        g = self._graphs.pop(key, None)  # (re-inserted below: dict order = recency)
        if g is None:  # first call of this shape: eager (and warm)
            self._graphs[key] = "warm"
            return fn(x)
        if isinstance(g, str):
            live = [k for k, v in self._graphs.items() if not isinstance(v, str)]
            while len(live) >= self.MAX_GRAPHS:  # LRU: the oldest captured shape makes room (its pool blocks go back to the shared pool)
                del self._graphs[live.pop(0)]
            static_in = x.clone()
            torch.cuda.synchronize()
            if getattr(self, "_graph_pool", None) is None:
                self._graph_pool = torch.cuda.graph_pool_handle()  # ONE pool for all VAE graphs: they never run concurrently, so a
            graph = torch.cuda.CUDAGraph()                          # dropped graph's activations are reused by the next capture
            with torch.cuda.graph(graph, pool=self._graph_pool):
                static_out = fn(static_in)
            g = (graph, static_in, static_out)
        self._graphs[key] = g
        graph, static_in, static_out = g
        static_in.copy_(x)
        graph.replay()
        return static_out.clone()

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        mu = self._run("encode", lambda t: torch.stack([self.engine().encode(t[b]) for b in range(t.shape[0])], 0), x).to(x.dtype)
        dist = SimpleNamespace(mode=lambda: mu, mean=mu)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        v = self._run("decode", lambda t: torch.stack([self.engine().decode(t[b]) for b in range(t.shape[0])], 0), z).to(z.dtype)
        return SimpleNamespace(sample=v) if return_dict else (v,)
