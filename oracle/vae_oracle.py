"""CPU oracle for the Wan-2.1 causal-3D-conv VAE (TEST INFRASTRUCTURE — never the product path).

Functional restatement (flat parameter dict, explicit frame caches) of the in-repo spec of diffusers
`AutoencoderKLWan`:  /root/reference/chronoedit/_src/tokenizers/wan2pt1.py
    CausalConv3d :42-60   RMS_norm :63-75   Resample :86-183   ResidualBlock :186-220   AttentionBlock :223-259
    Encoder3d :262-357    Decoder3d :360-457               WanVAE_.encode :502-541   .decode :543-560
Call sites on the hot path: chronoedit_diffusers/pipeline_chronoedit.py:442 (encode, mode of the posterior) and
:776-781 (decode).  Parameter names are the reference's (`encoder.downsamples.3.residual.2.weight`, ...).

Pinned by tests/golden/vae_*.pt, produced by executing the reference's own class definitions (lines 38-581 of that
file, exec'd without its un-importable module-level imports) on the same seeded weights: oracle/gen_golden_vae.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

CACHE_T = 2


@dataclass
class VAEConfig:
    """_video_vae cfg (wan2pt1.py:597-605): dim 96, mult [1,2,4,4], 2 res blocks, temporal down [F,T,T], z 16."""
    dim: int = 96
    z_dim: int = 16
    dim_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    temperal_downsample: Tuple[bool, ...] = (False, True, True)
    temporal_window: int = 4

    @property
    def temperal_upsample(self):
        return tuple(self.temperal_downsample[::-1])


LATENTS_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632,
                -0.1922, -0.9497, 0.2503, -0.2921]  # wan2pt1.py:697-714
LATENTS_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382,
               1.1253, 2.8251, 1.9160]  # wan2pt1.py:715-732


# ------------------------------------------------------------------------------------------
# architecture description: a flat list of layer specs, shared by shapes() and the forward code
# ------------------------------------------------------------------------------------------
def encoder_layers(cfg: VAEConfig):
    """('res', name, cin, cout) | ('down2d'|'down3d', name, c) | ('attn', name, c) in execution order (:283-305)."""
    dims = [cfg.dim * u for u in (1,) + tuple(cfg.dim_mult)]
    layers, idx = [], 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(cfg.num_res_blocks):
            layers.append(("res", f"encoder.downsamples.{idx}", cin, cout))
            idx += 1
            cin = cout
        if i != len(cfg.dim_mult) - 1:
            layers.append(("down3d" if cfg.temperal_downsample[i] else "down2d", f"encoder.downsamples.{idx}", cout))
            idx += 1
    c = dims[-1]
    mid = [("res", "encoder.middle.0", c, c), ("attn", "encoder.middle.1", c), ("res", "encoder.middle.2", c, c)]
    return layers, mid, c


def decoder_layers(cfg: VAEConfig):
    """(:384-415) note the halved in_dim after every upsample."""
    dims = [cfg.dim * u for u in (cfg.dim_mult[-1],) + tuple(cfg.dim_mult[::-1])]
    c0 = dims[0]
    mid = [("res", "decoder.middle.0", c0, c0), ("attn", "decoder.middle.1", c0), ("res", "decoder.middle.2", c0, c0)]
    layers, idx = [], 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(cfg.num_res_blocks + 1):
            layers.append(("res", f"decoder.upsamples.{idx}", cin, cout))
            idx += 1
            cin = cout
        if i != len(cfg.dim_mult) - 1:
            layers.append(("up3d" if cfg.temperal_upsample[i] else "up2d", f"decoder.upsamples.{idx}", cout))
            idx += 1
    return mid, layers, dims[-1]


def param_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}

    def conv3(name, cin, cout, k=(3, 3, 3)):
        s[name + ".weight"] = (cout, cin) + tuple(k)
        s[name + ".bias"] = (cout,)

    def res(name, cin, cout):
        s[name + ".residual.0.gamma"] = (cin, 1, 1, 1)
        conv3(name + ".residual.2", cin, cout)
        s[name + ".residual.3.gamma"] = (cout, 1, 1, 1)
        conv3(name + ".residual.6", cout, cout)
        if cin != cout:
            conv3(name + ".shortcut", cin, cout, (1, 1, 1))

    def attn(name, c):
        s[name + ".norm.gamma"] = (c, 1, 1)
        s[name + ".to_qkv.weight"] = (3 * c, c, 1, 1)
        s[name + ".to_qkv.bias"] = (3 * c,)
        s[name + ".proj.weight"] = (c, c, 1, 1)
        s[name + ".proj.bias"] = (c,)

    def layer(l):
        kind, name = l[0], l[1]
        if kind == "res":
            res(name, l[2], l[3])
        elif kind == "attn":
            attn(name, l[2])
        elif kind in ("down2d", "down3d"):
            c = l[2]
            s[name + ".resample.1.weight"] = (c, c, 3, 3)
            s[name + ".resample.1.bias"] = (c,)
            if kind == "down3d":
                conv3(name + ".time_conv", c, c, (3, 1, 1))
        elif kind in ("up2d", "up3d"):
            c = l[2]
            s[name + ".resample.1.weight"] = (c // 2, c, 3, 3)
            s[name + ".resample.1.bias"] = (c // 2,)
            if kind == "up3d":
                conv3(name + ".time_conv", c, 2 * c, (3, 1, 1))

    conv3("encoder.conv1", 3, cfg.dim)
    enc, emid, ec = encoder_layers(cfg)
    for l in enc + emid:
        layer(l)
    s["encoder.head.0.gamma"] = (ec, 1, 1, 1)
    conv3("encoder.head.2", ec, 2 * cfg.z_dim)
    conv3("conv1", 2 * cfg.z_dim, 2 * cfg.z_dim, (1, 1, 1))
    conv3("conv2", cfg.z_dim, cfg.z_dim, (1, 1, 1))
    dmid, dec, dc = decoder_layers(cfg)
    conv3("decoder.conv1", cfg.z_dim, dmid[0][2])
    for l in dmid + dec:
        layer(l)
    s["decoder.head.0.gamma"] = (dc, 1, 1, 1)
    conv3("decoder.head.2", dc, 3)
    return s


def make_synthetic_params(cfg: VAEConfig, seed: int = 4321, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights: convs ~ N(0, 1/fan_in) (keeps activations O(1) through ~60 layers), gammas 1 + 0.1 N."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("gamma"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) / fan_in**0.5
        out[name] = t.to(dtype)
    return out


# ------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------
def causal_conv3d(p, name, x, cache=None, stride=(1, 1, 1)):
    """CausalConv3d.forward (:52-60): time padding 2*pad_t in FRONT (taken from `cache` when given), symmetric in h, w."""
    w, b = p[name + ".weight"], p[name + ".bias"]
    kt, kh, kw = w.shape[2:]
    pt = (kt - 1) if stride[0] == 1 else 0  # the strided time_conv of downsample3d is built with padding 0 (:108)
    ph, pw = kh // 2, kw // 2
    if cache is not None and pt > 0:
        x = torch.cat([cache.to(x.device), x], dim=2)
        pt -= cache.shape[2]
    x = F.pad(x, (pw, pw, ph, ph, pt, 0))
    return F.conv3d(x, w, b, stride=stride)


def rms_norm(x, gamma, channel_dim=1):
    """RMS_norm (:63-75): F.normalize over channels * sqrt(C) * gamma."""
    return F.normalize(x, dim=channel_dim) * (x.shape[channel_dim] ** 0.5) * gamma


class Caches:
    """feat_cache list + running index of the reference (:571-580), one slot per CausalConv3d in execution order."""

    def __init__(self):
        self.slots: Dict[int, object] = {}
        self.idx = 0

    def begin(self):
        self.idx = 0

    def next(self):
        i = self.idx
        self.idx += 1
        return i


def _cached_conv(p, name, x, caches: Optional[Caches]):
    """The recurring cache idiom (:200-210, :311-316): keep the last 2 input frames (1 new + 1 old when the chunk has
    a single frame) for the next chunk, convolve with the previous chunk's cache."""
    if caches is None:
        return causal_conv3d(p, name, x)
    i = caches.next()
    prev = caches.slots.get(i)
    keep = x[:, :, -CACHE_T:].clone()
    if keep.shape[2] < 2 and prev is not None:
        keep = torch.cat([prev[:, :, -1:].to(keep.device), keep], dim=2)
    out = causal_conv3d(p, name, x, prev)
    caches.slots[i] = keep
    return out


def residual_block(p, name, x, caches):
    """ResidualBlock.forward (:205-220)."""
    h = causal_conv3d(p, name + ".shortcut", x) if (name + ".shortcut.weight") in p else x
    y = F.silu(rms_norm(x, p[name + ".residual.0.gamma"]))
    y = _cached_conv(p, name + ".residual.2", y, caches)
    y = F.silu(rms_norm(y, p[name + ".residual.3.gamma"]))
    y = _cached_conv(p, name + ".residual.6", y, caches)
    return y + h


def attention_block(p, name, x):
    """AttentionBlock.forward (:240-259): per-frame single-head attention over h*w."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = rms_norm(y, p[name + ".norm.gamma"])
    qkv = F.conv2d(y, p[name + ".to_qkv.weight"], p[name + ".to_qkv.bias"])
    q, k, v = qkv.reshape(b * t, 1, 3 * c, h * w).permute(0, 1, 3, 2).contiguous().chunk(3, dim=-1)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
    o = F.conv2d(o, p[name + ".proj.weight"], p[name + ".proj.bias"])
    o = o.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
    return o + x


def _per_frame_conv2d(p, name, x, stride=1, pad=(1, 1, 1, 1)):
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.conv2d(F.pad(y, pad), p[name + ".weight"], p[name + ".bias"], stride=stride)
    return y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def resample(p, kind, name, x, caches):
    """Resample.forward (:118-165)."""
    b, c, t, h, w = x.shape
    if kind == "up3d" and caches is not None:
        i = caches.next()
        prev = caches.slots.get(i)
        if prev is None:
            caches.slots[i] = "Rep"  # first chunk: no temporal upsampling of the very first frame (:121-124)
        else:
            keep = x[:, :, -CACHE_T:].clone()
            if keep.shape[2] < 2:
                if isinstance(prev, str):
                    keep = torch.cat([torch.zeros_like(keep), keep], dim=2)
                else:
                    keep = torch.cat([prev[:, :, -1:].to(keep.device), keep], dim=2)
            y = causal_conv3d(p, name + ".time_conv", x, None if isinstance(prev, str) else prev)
            caches.slots[i] = keep
            y = y.reshape(b, 2, c, t, h, w)
            x = torch.stack((y[:, 0], y[:, 1]), 3).reshape(b, c, t * 2, h, w)  # channel halves -> alternating frames
    if kind in ("up2d", "up3d"):
        bb, cc, tt, hh, ww = x.shape
        y = x.permute(0, 2, 1, 3, 4).reshape(bb * tt, cc, hh, ww)
        y = F.interpolate(y.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").type_as(y)
        y = F.conv2d(y, p[name + ".resample.1.weight"], p[name + ".resample.1.bias"], padding=1)
        x = y.reshape(bb, tt, *y.shape[1:]).permute(0, 2, 1, 3, 4)
    else:
        x = _per_frame_conv2d(p, name + ".resample.1", x, stride=2, pad=(0, 1, 0, 1))  # ZeroPad2d((0,1,0,1)) (:108-112)
        if kind == "down3d" and caches is not None:
            i = caches.next()
            prev = caches.slots.get(i)
            if prev is None:
                caches.slots[i] = x.clone()
            else:
                keep = x[:, :, -1:].clone()
                x = causal_conv3d(p, name + ".time_conv", torch.cat([prev[:, :, -1:], x], 2), stride=(2, 1, 1))
                caches.slots[i] = keep
    return x


def _run(p, layers, x, caches):
    for l in layers:
        if l[0] == "res":
            x = residual_block(p, l[1], x, caches)
        elif l[0] == "attn":
            x = attention_block(p, l[1], x)
        else:
            x = resample(p, l[0], l[1], x, caches)
    return x


def encoder_forward(p, cfg, x, caches):
    caches.begin()
    x = _cached_conv(p, "encoder.conv1", x, caches)
    enc, mid, c = encoder_layers(cfg)
    x = _run(p, enc, x, caches)
    x = _run(p, mid, x, caches)
    x = F.silu(rms_norm(x, p["encoder.head.0.gamma"]))
    return _cached_conv(p, "encoder.head.2", x, caches)


def decoder_forward(p, cfg, x, caches):
    caches.begin()
    x = _cached_conv(p, "decoder.conv1", x, caches)
    mid, dec, c = decoder_layers(cfg)
    x = _run(p, mid, x, caches)
    x = _run(p, dec, x, caches)
    x = F.silu(rms_norm(x, p["decoder.head.0.gamma"]))
    return _cached_conv(p, "decoder.head.2", x, caches)


def encode(p, cfg: VAEConfig, x: torch.Tensor) -> torch.Tensor:
    """WanVAE_.encode (:502-533) without the latent normalisation: chunks of 1, 4, 4, ... frames -> mu."""
    caches = Caches()
    t = x.shape[2]
    outs = [encoder_forward(p, cfg, x[:, :, :1], caches)]
    n = 1 + (t - 1) // cfg.temporal_window
    for i in range(1, n):
        outs.append(encoder_forward(p, cfg, x[:, :, 1 + cfg.temporal_window * (i - 1) : 1 + cfg.temporal_window * i], caches))
    if (t - 1) % cfg.temporal_window:
        outs.append(encoder_forward(p, cfg, x[:, :, 1 + cfg.temporal_window * (n - 1) :], caches))
    out = torch.cat(outs, 2)
    mu, _ = causal_conv3d(p, "conv1", out).chunk(2, dim=1)
    return mu


def decode(p, cfg: VAEConfig, z: torch.Tensor) -> torch.Tensor:
    """WanVAE_.decode (:543-560) without the latent de-normalisation: one latent frame at a time."""
    caches = Caches()
    x = causal_conv3d(p, "conv2", z)
    outs = [decoder_forward(p, cfg, x[:, :, i : i + 1], caches) for i in range(z.shape[2])]
    return torch.cat(outs, 2)
