"""torch-tensor front end of the C ABI (include/chronoedit_hip.h).

Every function checks device/dtype/contiguity, then hands raw device pointers and the current
torch stream to libchronoedit_hip.so.  There is no fallback: a CPU tensor or a missing library
raises.  PyTorch is used here only for device memory and streams.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import hiplib

EPI_BIAS, EPI_BIAS_GELU, EPI_GATE_RES, EPI_BIAS_GELU_ERF = 0, 1, 2, 3
EPI_F32, EPI_MUL, EPI_BIAS_ROW, EPI_BIAS_T = 4, 5, 6, 7

_lib = None      # the library launches go through: the product library unless a diagnostic selector is away from its default
_product = None

# The alternative kernel bodies are selectable only in libchronoedit_hip_diag.so (include/chronoedit_hip_diag.h): while every selector is at
# its default, launches go through the product library; the first non-default value moves them to the diagnostic build, the last reset moves
# them back.  Tools and body-equivalence tests only - the engine, the pipeline and bench.py's timed region never call set_*.
_KNOB_DEFAULTS = {"ce_set_gemm_variant": -1, "ce_set_attention_waves": 0, "ce_set_gemm_fp8_variant": 1,
                  "ce_set_attention_mxfp8_variant": 1, "ce_set_attention_mxfp8_persistent": 512}
_knobs = dict(_KNOB_DEFAULTS)


def lib():
    global _lib, _product
    if _lib is None:
        _lib = _product = hiplib.load()
        import os
        v8 = os.environ.get("CE_GEMM_FP8_VARIANT")  # A/B knobs for whole-step runs (bench.py under another main loop): the diagnostic build
        if v8 is not None:
            _set_knob("ce_set_gemm_fp8_variant", int(v8))
        v = os.environ.get("CE_GEMM_VARIANT")
        if v is not None:
            _set_knob("ce_set_gemm_variant", int(v))
    return _lib


_force_diag = False


def force_diagnostics(on: bool = True) -> None:
    """(tools) send launches through libchronoedit_hip_diag.so even with every selector at its default - e.g. to read the exact-route counters
    of the attention kernels (ce_diag_attention_exact_route_hits) for launches that run the DEFAULT bodies."""
    global _lib, _force_diag
    lib()
    _force_diag = bool(on)
    want = hiplib.load_diagnostics() if (_force_diag or any(_knobs[k] != _KNOB_DEFAULTS[k] for k in _knobs)) else _product
    if want is not _lib:
        _lib = want
        _gemm_ws.pop("active", None)


def attention_exact_route_hits(reset: bool = True):
    """(diagnostic build) (bf16 kernels, MXFP8 kernel): waves x key tiles that took the exact route of the speculative softmax since the last reset."""
    buf = (ctypes.c_ulonglong * 2)()
    _check(hiplib.load_diagnostics().ce_diag_attention_exact_route_hits(buf, 1 if reset else 0), "ce_diag_attention_exact_route_hits")
    return int(buf[0]), int(buf[1])


def _set_knob(symbol: str, v: int) -> int:
    global _lib
    lib()
    diag = hiplib.load_diagnostics()
    prev = getattr(diag, symbol)(int(v))
    _knobs[symbol] = getattr(diag, symbol)(int(v))  # (setters ignore values they do not know: read back what is in force)
    want = diag if (_force_diag or any(_knobs[k] != _KNOB_DEFAULTS[k] for k in _knobs)) else _product
    if want is not _lib:
        _lib = want
        _gemm_ws.pop("active", None)  # the split-K scratch registry is per library: re-register with the one launches now go through
    return prev


class HipKernelError(RuntimeError):
    pass


# ---- optional per-launch timing with HIP events on the launch stream (bench.py roofline leg) ----
_PROF = None  # list of (key, flops_or_bytes, start_event, end_event) while profiling


class profile:
    """with ops.profile() as prof: ...; prof.summary() -> {key: {n, total_ms, avg_ms, work}} (events on the
    current torch stream, which is the stream every launcher enqueues on)."""

    def __enter__(self):
        global _PROF
        _PROF = []
        return self

    def __exit__(self, *exc):
        global _PROF
        self.records, _PROF = _PROF, None
        return False

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, work, st, en in self.records:
            d = out.setdefault(key, {"n": 0, "total_ms": 0.0, "work": work})
            d["n"] += 1
            d["total_ms"] += st.elapsed_time(en)
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / d["n"]
        return out


def _prof_begin():
    if _PROF is None:
        return None
    st = torch.cuda.Event(enable_timing=True)
    st.record()
    return st


def _prof_end(st, key, work):
    if st is not None:
        en = torch.cuda.Event(enable_timing=True)
        en.record()
        _PROF.append((key, work, st, en))


def _check(rc: int, name: str):
    if rc != 0:
        kind = {-1: "bad argument", -2: "unsupported shape", -3: "misaligned stride"}.get(rc, f"hipError {rc}")
        raise HipKernelError(f"{name} failed: {kind}")


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _dev(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise HipKernelError(f"{name}: tensor must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def _rows(t: torch.Tensor, name: str):
    """2-D view with unit inner stride -> (M, D, ld)."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a 2-D tensor with contiguous rows, got shape {tuple(t.shape)} stride {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


def ln_affine(x: torch.Tensor, a: torch.Tensor, b: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None,
              ab_rows: int = 0, ab_stride: int = 0):
    """out = LN_fp32(x) * a + b  (bf16 in/out, fp32 a/b).  ab_rows > 0: row m uses a/b + (m // ab_rows) * ab_stride."""
    _dev(x, torch.bfloat16, "x"), _dev(a, torch.float32, "a"), _dev(b, torch.float32, "b")
    M, D, ldx = _rows(x, "x")
    if out is None:
        out = torch.empty((M, D), dtype=torch.bfloat16, device=x.device)
    _, _, ldy = _rows(out, "out")
    st = _prof_begin()
    _check(lib().ce_ln_affine_bf16(_ptr(x), _ptr(out), _ptr(a), _ptr(b), M, D, ldx, ldy, float(eps), int(ab_rows), int(ab_stride),
                                   _stream()), "ce_ln_affine_bf16")
    _prof_end(st, f"ln_affine_{M}x{D}", 4.0 * M * D)
    return out


def rmsnorm_rope_(x: torch.Tensor, w: torch.Tensor, cos_sin: Optional[torch.Tensor], head_dim: int, eps: float,
                  x2: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None):
    """In place RMSNorm-across-heads (+ RoPE when cos_sin [R, head_dim/2, 2] fp32 is given; row m uses entry m % R).
    (x2, w2): a second tensor of the same geometry (k beside q in the fused qkv buffer) handled by the same launch."""
    _dev(x, torch.bfloat16, "x"), _dev(w, torch.float32, "w")
    M, D, ld = _rows(x, "x")
    rope_rows = 0
    if cos_sin is not None:
        _dev(cos_sin, torch.float32, "cos_sin")
        assert cos_sin.is_contiguous() and cos_sin.shape[1:] == (head_dim // 2, 2) and M % cos_sin.shape[0] == 0, (cos_sin.shape, M)
        rope_rows = cos_sin.shape[0]
    if x2 is not None:
        _dev(x2, torch.bfloat16, "x2"), _dev(w2, torch.float32, "w2")
        assert _rows(x2, "x2") == (M, D, ld)
    st = _prof_begin()
    _check(lib().ce_rmsnorm_rope_bf16(_ptr(x), _ptr(w), _ptr(x2), _ptr(w2), _ptr(cos_sin), M, D, ld, head_dim, float(eps), rope_rows,
                                      _stream()), "ce_rmsnorm_rope_bf16")
    _prof_end(st, f"rmsnorm_rope_{M}x{D}" + ("x2" if x2 is not None else ""), (8.0 if x2 is not None else 4.0) * M * D)
    return x


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: Optional[torch.Tensor] = None,
         epilogue: int = EPI_BIAS, gate: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, gate_rows: int = 0):
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T + bias).  gate [N] or, with gate_rows > 0, [M/gate_rows, N] (one per sample).
    a may be a 3-D [S, M, K/S] tensor (contiguous): the K-segmented operand an all-to-all leaves behind (ce_gemm_aseg_bf16).
    epilogue EPI_BIAS_T: out is [N, M] and receives the TRANSPOSE of a @ w^T + bias[n] (CE_EPI_BIAS_T)."""
    _dev(a, torch.bfloat16, "a"), _dev(w, torch.bfloat16, "w")
    a_seg_k, a_seg_stride = 0, 0
    if a.dim() == 3:
        S, M, Ks = a.shape
        if a.stride(2) != 1 or a.stride(1) < Ks:
            raise ValueError(f"gemm: segmented a needs unit inner stride, got {a.stride()}")
        K, lda, a_seg_k, a_seg_stride = S * Ks, a.stride(1), Ks, a.stride(0)
    else:
        M, K, lda = _rows(a, "a")
    N, K2, ldw = _rows(w, "w")
    if K != K2:
        raise ValueError(f"gemm: K mismatch {K} vs {K2}")
    if bias is not None:
        _dev(bias, torch.float32, "bias")
        assert bias.numel() == (M if epilogue == EPI_BIAS_ROW else N) and bias.is_contiguous()
    oshape = (N, M) if epilogue == EPI_BIAS_T else (M, N)
    if out is None:
        out = torch.empty(oshape, dtype=torch.bfloat16, device=a.device)
    _dev(out, torch.bfloat16, "out")
    Mo, No, ldc = _rows(out, "out")
    assert (Mo, No) == oshape, ((Mo, No), oshape)
    ldres = 0
    if epilogue in (EPI_GATE_RES, EPI_MUL):
        if res is None:
            raise ValueError("EPI_GATE_RES / EPI_MUL need res")
        _dev(res, torch.bfloat16, "res")
        _, _, ldres = _rows(res, "res")
        if gate is not None:
            _dev(gate, torch.float32, "gate")
            assert gate.is_contiguous() and gate.numel() == (N if gate_rows <= 0 else (M + gate_rows - 1) // gate_rows * N)
    ensure_gemm_workspace(a.device)
    st = _prof_begin()
    _check(lib().ce_gemm_aseg_bf16(_ptr(a), _ptr(w), _ptr(out), _ptr(bias), epilogue, _ptr(gate), _ptr(res), M, N, K, lda, ldw, ldc,
                                   ldres, int(gate_rows), int(a_seg_k), int(a_seg_stride), _stream()), "ce_gemm_bf16")
    _prof_end(st, f"gemm_{M}x{N}x{K}_epi{epilogue}", 2.0 * M * N * K)
    return out


def rope_scatter(x: torch.Tensor, cols, weights, D: int, world: int, cos_sin: Optional[torch.Tensor], head_dim: int, eps: float,
                 out: Optional[torch.Tensor] = None):
    """Ulysses send buffer [world, M, len(cols), D/world] from column blocks `cols` (start columns, each D wide) of x [M, ld]:
    RMSNorm * weights[i] (+ RoPE with cos_sin) where weights[i] is given, plain copy where it is None (ce_rope_scatter_bf16)."""
    _dev(x, torch.bfloat16, "x")
    M, _, ldx = _rows(x, "x")
    nt = len(cols)
    assert 1 <= nt <= 3 and len(weights) == nt
    for w in weights:
        if w is not None:
            _dev(w, torch.float32, "w")
            assert w.is_contiguous() and w.numel() == D
    rope_rows = 0
    if cos_sin is not None:
        _dev(cos_sin, torch.float32, "cos_sin")
        assert cos_sin.is_contiguous() and cos_sin.shape[1:] == (head_dim // 2, 2) and M % cos_sin.shape[0] == 0
        rope_rows = cos_sin.shape[0]
    if out is None:
        out = torch.empty((world, M, nt, D // world), dtype=torch.bfloat16, device=x.device)
    _dev(out, torch.bfloat16, "out")
    assert out.is_contiguous() and out.numel() == M * nt * D
    c = list(cols) + [0] * (3 - nt)
    w = list(weights) + [None] * (3 - nt)
    st = _prof_begin()
    _check(lib().ce_rope_scatter_bf16(_ptr(x), ldx, _ptr(out), M, D, world, nt, c[0], _ptr(w[0]), c[1], _ptr(w[1]), c[2], _ptr(w[2]),
                                      _ptr(cos_sin), head_dim, float(eps), rope_rows, _stream()), "ce_rope_scatter_bf16")
    _prof_end(st, f"rope_scatter_{M}x{D}x{nt}", 4.0 * M * D * nt)
    return out


_gemm_ws = {}
_gemm_split = True
GEMM_WS_BYTES = 256 * 384 * 256 * 4  # one fp32 slab of the LARGEST macro tile (384 x 256) per CU: any split-K tail the dispatcher may choose fits (96 MiB; round 6 - with 64 MiB the
# 384-row tile could not cut its tail at M = 7 200 and lost FFN-down by 13 %: profiles/r06_gemm_tile_choice_m7200.txt)


GEMM_WS_STREAMS = 8  # streams per device that get a scratch of their own; further ones share the device default


def ensure_gemm_workspace(device: torch.device) -> None:
    """Hand the library its split-K scratch: one default buffer per device (allocated once, outside any graph capture) and, for every
    further stream GEMMs are launched on, a buffer of that stream's own (ce_set_gemm_workspace_stream) - concurrent streams never share
    slabs.  A stream that is capturing a graph uses the default (nothing is allocated under capture; captures of one device do not
    overlap in this package)."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev, _gemm_split, stream)
    if _gemm_ws.get("active") == key:
        return
    if _gemm_split:
        buf = _gemm_ws.get(dev)
        first = _gemm_ws.get(("first_stream", dev))
        if buf is None:
            buf = torch.empty(GEMM_WS_BYTES, dtype=torch.uint8, device=device)
            _gemm_ws[dev] = buf
            _gemm_ws[("first_stream", dev)] = first = stream
        with torch.cuda.device(dev):
            _check(lib().ce_set_gemm_workspace(buf.data_ptr(), buf.numel()), "ce_set_gemm_workspace")
            if stream != first and (dev, stream) in _gemm_ws:
                _gemm_ws[(dev, stream)] = _gemm_ws.pop((dev, stream))  # most recently used last (dicts keep insertion order)
            elif stream != first and not torch.cuda.is_current_stream_capturing():
                mine = [k for k in _gemm_ws if isinstance(k, tuple) and len(k) == 2 and k[0] == dev]  # (dev, stream) keys, least recently used first
                if len(mine) >= GEMM_WS_STREAMS:
                    # the cap is reached: the LEAST RECENTLY USED stream is unregistered (it falls back to the device default if it ever comes
                    # back) and its scratch is DROPPED, not handed on: split-K partials of GEMMs still queued on that stream may be in it, and
                    # nothing orders the new stream behind them.  The caching allocator returns the block to the pool of the stream it was
                    # allocated on, so whatever reuses it is ordered behind that stream's queued work.
                    oldest = mine[0]
                    del _gemm_ws[oldest]
                    _check(lib().ce_set_gemm_workspace_stream(oldest[1], None, 0), "ce_set_gemm_workspace_stream")
                own = torch.empty(GEMM_WS_BYTES, dtype=torch.uint8, device=device)
                _gemm_ws[(dev, stream)] = own
                _check(lib().ce_set_gemm_workspace_stream(stream, own.data_ptr(), own.numel()), "ce_set_gemm_workspace_stream")
    else:
        with torch.cuda.device(dev):
            _check(lib().ce_set_gemm_workspace(None, 0), "ce_set_gemm_workspace")
            for k in [k for k in _gemm_ws if isinstance(k, tuple) and len(k) == 2 and k[0] == dev]:
                _check(lib().ce_set_gemm_workspace_stream(k[1], None, 0), "ce_set_gemm_workspace_stream")
                del _gemm_ws[k]
    _gemm_ws["active"] = key


def set_gemm_split(on: bool) -> bool:
    """Enable/disable the split-K tail of the 256-tile GEMM; returns the previous setting."""
    global _gemm_split
    prev, _gemm_split = _gemm_split, bool(on)
    return prev


def set_gemm_variant(v: int) -> int:
    """-1 auto, 0 force the 128-tile kernel, 1 force the 256-tile LDS-DMA kernel; returns the previous setting."""
    return _set_knob("ce_set_gemm_variant", v)


def set_attention_waves(n: int) -> int:
    """Attention loop body: 0 auto (= 64), 4 / 8 plain kernel with that many waves, 64 software-pipelined (default), 128 / 129 the
    one-wave-per-SIMD body of the V^T form (one workgroup per item / persistent); returns the previous value (include/chronoedit_hip.h)."""
    return _set_knob("ce_set_attention_waves", n)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, out: Optional[torch.Tensor] = None,
              k2: Optional[torch.Tensor] = None, v2: Optional[torch.Tensor] = None, scale: Optional[float] = None,
              batch: int = 1):
    """q [Nq, H*128], k/v [Nkv, H*128] (row strides free) -> out [Nq, H*128]; optional 2nd kv segment.
    batch > 1: every operand holds `batch` samples stacked along its rows (q [batch*Nq, ...], k [batch*Nkv, ...])."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        _dev(t, torch.bfloat16, n)
    Nq, Dq, ldq = _rows(q, "q")
    head_dim = Dq // heads
    L1, _, ldk = _rows(k, "k")
    L1v, _, ldv = _rows(v, "v")
    assert L1 == L1v
    if batch < 1 or Nq % batch or L1 % batch or (k2 is not None and k2.shape[0] % batch):
        raise ValueError(f"attention: row counts must be multiples of batch={batch}")
    if out is None:
        out = torch.empty((Nq, Dq), dtype=torch.bfloat16, device=q.device)
    _, _, ldo = _rows(out, "out")
    L2 = ldk2 = ldv2 = 0
    if k2 is not None:
        _dev(k2, torch.bfloat16, "k2"), _dev(v2, torch.bfloat16, "v2")
        L2, _, ldk2 = _rows(k2, "k2")
        _, _, ldv2 = _rows(v2, "v2")
    if scale is None:
        scale = head_dim ** -0.5
    st = _prof_begin()
    nq, l1, l2 = Nq // batch, L1 // batch, L2 // batch
    _check(lib().ce_attention_batched_bf16(_ptr(q), _ptr(k), _ptr(v), l1, ldk, ldv, _ptr(k2), _ptr(v2), l2, ldk2, ldv2, _ptr(out),
                                           nq, heads, head_dim, ldq, ldo, float(scale), batch, _stream()), "ce_attention_bf16")
    _prof_end(st, f"attention_{nq}x{l1}+{l2}_h{heads}" + (f"_b{batch}" if batch > 1 else ""),
              4.0 * nq * (l1 + l2) * head_dim * heads * batch)
    return out


def vt_columns(n_keys_total: int) -> int:
    """Row length of the V^T operand for `n_keys_total` keys (all samples side by side): whole 64-key strips, 16-B aligned rows."""
    return (n_keys_total + 63) // 64 * 64 + 64


def v_transpose(v: torch.Tensor, heads: int, out: Optional[torch.Tensor] = None):
    """v [keys of all samples, heads*128] (row stride free) -> V^T [heads*128, vt_columns(keys)] bf16, padding columns zeroed:
    the V operand of `attention_vt`."""
    _dev(v, torch.bfloat16, "v")
    n, D, ldv = _rows(v, "v")
    assert D == heads * 128
    if out is None:
        out = torch.empty((D, vt_columns(n)), dtype=torch.bfloat16, device=v.device)
    _dev(out, torch.bfloat16, "vt")
    assert out.shape[0] == D and out.is_contiguous() and out.shape[1] >= n
    st = _prof_begin()
    _check(lib().ce_v_transpose_bf16(_ptr(v), ldv, _ptr(out), out.shape[1], n, heads, _stream()), "ce_v_transpose_bf16")
    _prof_end(st, f"v_transpose_{n}x{D}", 4.0 * n * D)
    return out


def attention_vt(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, out: Optional[torch.Tensor] = None,
                 scale: Optional[float] = None, batch: int = 1):
    """Self-attention with V transposed (`v_transpose`): q [batch*Nq, H*128], k [batch*Nkv, H*128], vt [H*128, >= batch*Nkv (+pad)];
    K and V^T tiles both reach LDS by LDS-DMA.  Same arithmetic as `attention` (one KV segment)."""
    for n, t in (("q", q), ("k", k), ("vt", vt)):
        _dev(t, torch.bfloat16, n)
    Nq, Dq, ldq = _rows(q, "q")
    L1, _, ldk = _rows(k, "k")
    assert Dq == heads * 128 and vt.shape[0] == Dq and vt.stride(1) == 1
    if batch < 1 or Nq % batch or L1 % batch:
        raise ValueError(f"attention_vt: row counts must be multiples of batch={batch}")
    if out is None:
        out = torch.empty((Nq, Dq), dtype=torch.bfloat16, device=q.device)
    _, _, ldo = _rows(out, "out")
    if scale is None:
        scale = 128 ** -0.5
    st = _prof_begin()
    nq, l1 = Nq // batch, L1 // batch
    _check(lib().ce_attention_vt_bf16(_ptr(q), _ptr(k), _ptr(vt), l1, ldk, vt.stride(0), _ptr(out), nq, heads, 128, ldq, ldo, float(scale),
                                      batch, _stream()), "ce_attention_vt_bf16")
    _prof_end(st, f"attention_{nq}x{l1}+0_h{heads}" + (f"_b{batch}" if batch > 1 else ""), 4.0 * nq * l1 * 128 * heads * batch)
    return out


def attention_2seg_vt(q: torch.Tensor, k1: torch.Tensor, v1t: torch.Tensor, len1: int, k2: torch.Tensor, v2t: torch.Tensor, len2: int,
                      heads: int, out: Optional[torch.Tensor] = None, scale: Optional[float] = None, batch: int = 1,
                      cols1: Optional[int] = None, cols2: Optional[int] = None, out8: Optional[torch.Tensor] = None,
                      scale8: Optional[torch.Tensor] = None):
    """Cross-attention over two key / value segments (a softmax each, outputs added in bf16) with V transposed: k1 [batch*len1, H*128],
    v1t [H*128, >= (batch-1)*c1 + 64*ceil(len1/64)] with sample b's keys at columns [b*c1, b*c1 + len1), c1 = cols1 (default
    v1t.shape[1] // batch; even, >= len1; every column finite); the same for segment 2.  Same arithmetic as `attention(..., k2=, v2=)`; K and V^T tiles by LDS-DMA."""
    for n, t in (("q", q), ("k1", k1), ("v1t", v1t), ("k2", k2), ("v2t", v2t)):
        _dev(t, torch.bfloat16, n)
    Nq, Dq, ldq = _rows(q, "q")
    _, _, ldk1 = _rows(k1, "k1")
    _, _, ldk2 = _rows(k2, "k2")
    assert Dq == heads * 128 and v1t.shape[0] == Dq and v2t.shape[0] == Dq and v1t.stride(1) == 1 and v2t.stride(1) == 1
    assert Nq % batch == 0 and k1.shape[0] == batch * len1 and k2.shape[0] == batch * len2
    if cols1 is None:  # column stride between samples: explicit, or the row length split evenly
        assert v1t.shape[1] % batch == 0
        cols1 = v1t.shape[1] // batch
    if cols2 is None:
        assert v2t.shape[1] % batch == 0
        cols2 = v2t.shape[1] // batch
    if scale is None:
        scale = 128 ** -0.5
    nq = Nq // batch
    if out8 is not None:  # the result as the MX fp8 operand of the out-projection (e4m3 rows + tiled E8M0 block scales): no bf16 output
        _dev(out8, torch.uint8, "out8"), _dev(scale8, torch.uint8, "scale8")
        _, D8, ld8 = _rows(out8, "out8")
        assert out8.shape[0] == Nq and D8 == Dq and scale8.is_contiguous() and scale8.numel() >= mx_scale_bytes(Nq, Dq)
        st = _prof_begin()
        _check(lib().ce_attention_2seg_vt_quant_bf16(_ptr(q), _ptr(k1), _ptr(v1t), len1, ldk1, v1t.stride(0), int(cols1), _ptr(k2), _ptr(v2t),
                                                     len2, ldk2, v2t.stride(0), int(cols2), _ptr(out8), _ptr(scale8), nq, heads, 128, ldq, ld8,
                                                     float(scale), batch, _stream()), "ce_attention_2seg_vt_quant_bf16")
        _prof_end(st, f"attention_{nq}x{len1}+{len2}_h{heads}" + (f"_b{batch}" if batch > 1 else "") + "_mxq",
                  4.0 * nq * (len1 + len2) * 128 * heads * batch)
        return out8
    if out is None:
        out = torch.empty((Nq, Dq), dtype=torch.bfloat16, device=q.device)
    _, _, ldo = _rows(out, "out")
    st = _prof_begin()
    _check(lib().ce_attention_2seg_vt_bf16(_ptr(q), _ptr(k1), _ptr(v1t), len1, ldk1, v1t.stride(0), int(cols1), _ptr(k2), _ptr(v2t),
                                           len2, ldk2, v2t.stride(0), int(cols2), _ptr(out), nq, heads, 128, ldq, ldo,
                                           float(scale), batch, _stream()), "ce_attention_2seg_vt_bf16")
    _prof_end(st, f"attention_{nq}x{len1}+{len2}_h{heads}" + (f"_b{batch}" if batch > 1 else ""), 4.0 * nq * (len1 + len2) * 128 * heads * batch)
    return out


def v_transpose_blocked(v: torch.Tensor, heads: int, batch: int, blk_rows: int, n_keys: int, out: Optional[torch.Tensor] = None):
    """v [W * batch * blk_rows, heads*128] in the all-to-all receive layout ([source rank][sample][local token]) -> V^T
    [heads*128, batch * vt_sample_cols] plain per sample (`attention_vt_blocked`); n_keys = valid tokens per sample."""
    _dev(v, torch.bfloat16, "v")
    rows, D, ldv = _rows(v, "v")
    assert D == heads * 128 and blk_rows % 64 == 0 and rows % (batch * blk_rows) == 0
    cols = (n_keys + 63) // 64 * 64 + 64
    if out is None:
        out = torch.empty((D, batch * cols), dtype=torch.bfloat16, device=v.device)
    assert out.shape == (D, batch * cols) and out.is_contiguous()
    st = _prof_begin()
    _check(lib().ce_v_transpose_blocked_bf16(_ptr(v), ldv, _ptr(out), out.shape[1], n_keys, heads, batch, blk_rows, batch * blk_rows, cols,
                                             _stream()), "ce_v_transpose_blocked_bf16")
    _prof_end(st, f"v_transpose_{batch * n_keys}x{D}", 4.0 * batch * n_keys * D)
    return out


def attention_vt_blocked(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, batch: int, blk_rows: int, n_keys: int,
                         out: Optional[torch.Tensor] = None, scale: Optional[float] = None):
    """`attention_vt` over the blocked row layout: q / k / out [W * batch * blk_rows, heads*128] with token g of sample b in row
    (g // blk_rows) * batch * blk_rows + b * blk_rows + g % blk_rows; vt from `v_transpose_blocked`; n_keys valid keys per sample."""
    for n, t in (("q", q), ("k", k), ("vt", vt)):
        _dev(t, torch.bfloat16, n)
    rows, Dq, ldq = _rows(q, "q")
    rk, _, ldk = _rows(k, "k")
    assert Dq == heads * 128 and vt.shape[0] == Dq and vt.stride(1) == 1 and rows == rk and rows % (batch * blk_rows) == 0
    nq = rows // batch  # query tokens per sample = W * blk_rows
    cols = vt.shape[1] // batch
    if out is None:
        out = torch.empty((rows, Dq), dtype=torch.bfloat16, device=q.device)
    _, _, ldo = _rows(out, "out")
    if scale is None:
        scale = 128 ** -0.5
    st = _prof_begin()
    _check(lib().ce_attention_vt_blocked_bf16(_ptr(q), _ptr(k), _ptr(vt), n_keys, ldk, vt.stride(0), _ptr(out), nq, heads, 128, ldq, ldo,
                                              float(scale), batch, blk_rows, batch * blk_rows, cols, _stream()), "ce_attention_vt_blocked_bf16")
    _prof_end(st, f"attention_{nq}x{n_keys}+0_h{heads}" + (f"_b{batch}" if batch > 1 else ""), 4.0 * nq * n_keys * 128 * heads * batch)
    return out


def timestep_sinusoid(t: torch.Tensor, dim: int, out: Optional[torch.Tensor] = None):
    """[cos, sin] table of one timestep: int64 (diffusers path) or float32 (the sibling stacks' float timesteps)."""
    if t.dtype == torch.float32:
        _dev(t, torch.float32, "timestep")
    else:
        _dev(t, torch.int64, "timestep")
    if out is None:
        out = torch.empty((dim,), dtype=torch.float32, device=t.device)
    if t.dtype == torch.float32:
        _check(lib().ce_timestep_sinusoid_f32(_ptr(t), _ptr(out), dim, _stream()), "ce_timestep_sinusoid_f32")
    else:
        _check(lib().ce_timestep_sinusoid(_ptr(t), _ptr(out), dim, _stream()), "ce_timestep_sinusoid")
    return out


def gemv(w: torch.Tensor, x: torch.Tensor, bias: Optional[torch.Tensor], flags: int = 0, out: Optional[torch.Tensor] = None):
    """y = post(w @ pre(x) + bias); x/y fp32, w fp32 or bf16 (see include/chronoedit_hip.h for flags)."""
    _dev(x, torch.float32, "x")
    assert w.is_cuda and w.dtype in (torch.float32, torch.bfloat16) and w.is_contiguous()
    N, K = w.shape
    assert x.numel() == K
    if bias is not None:
        _dev(bias, torch.float32, "bias")
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
    _check(lib().ce_gemv(_ptr(w), int(w.dtype == torch.bfloat16), _ptr(x), _ptr(bias), _ptr(out), N, K, flags, _stream()), "ce_gemv")
    return out


def modulation(table: torch.Tensor, v: torch.Tensor, one_mask: int, out: Optional[torch.Tensor] = None):
    """out[L,J,D] = table[L,J,D] + v[J or 1, D]  (+1 on rows in one_mask)."""
    _dev(table, torch.float32, "table"), _dev(v, torch.float32, "v")
    L, J, D = table.shape
    v_rows = v.numel() // D
    assert table.is_contiguous() and v.is_contiguous() and v_rows in (1, J)
    if out is None:
        out = torch.empty_like(table)
    _check(lib().ce_modulation(_ptr(table), _ptr(v), _ptr(out), L, J, D, v_rows, one_mask, _stream()), "ce_modulation")
    return out


def patchify(x: torch.Tensor, kpad: int, out: Optional[torch.Tensor] = None, row0: int = 0, nrows: Optional[int] = None):
    """x [C,T,H,W] bf16 -> [T*(H/2)*(W/2), kpad]; (row0, nrows): only that range of token rows (zero rows past the last token)."""
    _dev(x, torch.bfloat16, "x")
    assert x.is_contiguous() and x.dim() == 4
    C, T, H, W = x.shape
    n = T * (H // 2) * (W // 2) if nrows is None else int(nrows)
    if out is None:
        out = torch.empty((n, kpad), dtype=torch.bfloat16, device=x.device)
    assert out.is_contiguous() and out.shape == (n, kpad)
    _check(lib().ce_patchify_rows_bf16(_ptr(x), _ptr(out), C, T, H, W, kpad, int(row0), n, _stream()), "ce_patchify_rows_bf16")
    return out


def unpatchify(y: torch.Tensor, cout: int, T: int, H: int, W: int, out: Optional[torch.Tensor] = None):
    _dev(y, torch.bfloat16, "y")
    _, _, ldy = _rows(y, "y")
    if out is None:
        out = torch.empty((cout, T, H, W), dtype=torch.bfloat16, device=y.device)
    _check(lib().ce_unpatchify_bf16(_ptr(y), _ptr(out), cout, T, H, W, ldy, _stream()), "ce_unpatchify_bf16")
    return out


def cfg_unipc_step(v_cond: torch.Tensor, v_uncond: Optional[torch.Tensor], x: torch.Tensor, x_last: torch.Tensor, m0: torch.Tensor,
                   m1: torch.Tensor, coef: torch.Tensor, x0_out: Optional[torch.Tensor] = None, round_sigma_v: bool = True,
                   bf16_state: bool = False):
    """Fused CFG + flow-UniPC update, in place on (x, x_last, m0, m1).  coef = device float[10].  bf16_state: the stored
    latents / history carry bf16 values like the reference's bf16 tensors (fp32 storage)."""
    _dev(v_cond, torch.bfloat16, "v_cond")
    for n, t in (("x", x), ("x_last", x_last), ("m0", m0), ("m1", m1), ("coef", coef)):
        _dev(t, torch.float32, n)
        assert t.is_contiguous()
    if v_uncond is not None:
        _dev(v_uncond, torch.bfloat16, "v_uncond")
        assert v_uncond.is_contiguous()
    assert v_cond.is_contiguous() and coef.numel() >= 10
    n = x.numel()
    assert v_cond.numel() == n == x_last.numel() == m0.numel() == m1.numel()
    _check(lib().ce_cfg_unipc_step(_ptr(v_cond), _ptr(v_uncond), _ptr(x), _ptr(x_last), _ptr(m0), _ptr(m1), _ptr(x0_out), _ptr(coef),
                                   _ptr(None), n, int(round_sigma_v) | (2 if bf16_state else 0), _stream()), "ce_cfg_unipc_step")
    return x


# ---- Wan VAE ------------------------------------------------------------------------------------------------------
def conv3d_gemm(in_stack: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], out_stack: torch.Tensor,
                res_stack: Optional[torch.Tensor], *, T_out: int, H: int, W: int, Cin: int, Cout: int, KT: int, n_tile: int = 0):
    """Stride-1 3x3 / 3x3x3 conv of a wide layer as ONE large-tile GEMM over a contiguous stack of bordered frames (see
    include/chronoedit_hip.h: in_stack holds T_out + KT - 1 frames plus one zeroed slack frame)."""
    _dev(in_stack, torch.bfloat16, "in_stack")
    _dev(out_stack, torch.bfloat16, "out_stack")
    _dev(weight, torch.bfloat16, "weight")
    assert in_stack.is_contiguous() and out_stack.is_contiguous() and weight.is_contiguous() and weight.dim() == 2
    assert in_stack.shape[0] >= T_out + KT and tuple(in_stack.shape[1:]) == (H + 2, W + 2, Cin), in_stack.shape
    assert out_stack.shape[0] >= T_out and tuple(out_stack.shape[1:3]) == (H + 2, W + 2), out_stack.shape
    if res_stack is not None:
        _dev(res_stack, torch.bfloat16, "res_stack")
        assert res_stack.is_contiguous() and res_stack.shape[1:] == out_stack.shape[1:] and res_stack.shape[0] >= T_out
    st_ev = _prof_begin()
    _check(lib().ce_conv3d_gemm_bf16(_ptr(in_stack), _ptr(weight), weight.shape[1], _ptr(bias), _ptr(out_stack), _ptr(res_stack), T_out, H, W,
                                     Cin, Cout, KT, out_stack.shape[3], n_tile, _stream()), "ce_conv3d_gemm_bf16")
    _prof_end(st_ev, f"conv_{KT}x3x3_{Cin}->{Cout}_{T_out}x{H}x{W}", 2.0 * T_out * H * W * Cout * Cin * KT * 9)


def conv3d_gemm_rms_silu(in_stack: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], normed_stack: torch.Tensor, gamma: torch.Tensor, *,
                         T_out: int, H: int, W: int, Cin: int, Cout: int, KT: int, silu: bool = True, out_stack: Optional[torch.Tensor] = None,
                         res_stack: Optional[torch.Tensor] = None):
    """conv3d_gemm with the next layer's RMS_norm (+ SiLU) in its epilogue: normed_stack = silu(rms_norm(y) * gamma), y = bf16(conv + bias)
    [bf16(res + .)]; y itself goes to out_stack when one is given (the next block's shortcut operand), else it is never written.  96 output
    channels only (ce_conv3d_gemm_rms_silu_bf16)."""
    _dev(in_stack, torch.bfloat16, "in_stack")
    _dev(normed_stack, torch.bfloat16, "normed_stack")
    _dev(weight, torch.bfloat16, "weight")
    _dev(gamma, torch.float32, "gamma")
    assert in_stack.is_contiguous() and normed_stack.is_contiguous() and weight.is_contiguous() and weight.dim() == 2 and gamma.numel() == Cout
    assert in_stack.shape[0] >= T_out + KT and tuple(in_stack.shape[1:]) == (H + 2, W + 2, Cin), in_stack.shape
    assert normed_stack.shape[0] >= T_out and tuple(normed_stack.shape[1:3]) == (H + 2, W + 2), normed_stack.shape
    for t, nm in ((out_stack, "out_stack"), (res_stack, "res_stack")):
        if t is not None:
            _dev(t, torch.bfloat16, nm)
            assert t.is_contiguous() and t.shape[1:] == normed_stack.shape[1:] and t.shape[0] >= T_out, (nm, t.shape)
    assert res_stack is None or out_stack is not None
    st_ev = _prof_begin()
    _check(lib().ce_conv3d_gemm_rms_silu_bf16(_ptr(in_stack), _ptr(weight), weight.shape[1], _ptr(bias), _ptr(out_stack), _ptr(res_stack), _ptr(normed_stack),
                                              T_out, H, W, Cin, Cout, KT, normed_stack.shape[3], _ptr(gamma), int(bool(silu)), _stream()),
           "ce_conv3d_gemm_rms_silu_bf16")
    _prof_end(st_ev, f"conv_{KT}x3x3_{Cin}->{Cout}_{T_out}x{H}x{W}+norm", 2.0 * T_out * H * W * Cout * Cin * KT * 9)


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def conv_igemm(in_frames, weight: torch.Tensor, bias: Optional[torch.Tensor], out_frames, res_frames, *, Cin: int, Cout: int,
               KT: int, KH: int, KW: int, st: int, ss: int, H_out: int, W_out: int, in_Wp: int, in_off: int, out_Wp: int,
               out_border: int, out_cstride: int, out_coff: int = 0):
    """Implicit-GEMM conv over lists of channels-last frames (see include/chronoedit_hip.h)."""
    for t in list(in_frames) + list(out_frames) + list(res_frames or []):
        _dev(t, torch.bfloat16, "frame")
    _dev(weight, torch.bfloat16, "weight")
    assert weight.is_contiguous() and weight.numel() >= Cout * KT * KH * KW * Cin
    ia, oa = _ptr_array(in_frames), _ptr_array(out_frames)
    ra = _ptr_array(res_frames) if res_frames else None
    st_ev = _prof_begin()
    _check(lib().ce_conv_igemm_bf16(ia, len(in_frames), _ptr(weight), _ptr(bias), oa, len(out_frames), ra, Cin, Cout, KT, KH, KW,
                                    st, ss, H_out, W_out, in_Wp, in_off, in_off, out_Wp, out_border, out_cstride, out_coff,
                                    _stream()), "ce_conv_igemm_bf16")
    _prof_end(st_ev, f"conv_{KT}x{KH}x{KW}_{Cin}->{Cout}_{len(out_frames)}x{H_out}x{W_out}",
              2.0 * len(out_frames) * H_out * W_out * Cout * Cin * KT * KH * KW)


def conv3d_head(in_frames, weight: torch.Tensor, bias: Optional[torch.Tensor], out_frames, *, Cin: int, Cout: int, KT: int, H_out: int,
                W_out: int, in_Wp: int, out_Wp: int, out_border: int, out_cstride: int, out_coff: int = 0):
    """Stride-1 3x3(x3) conv of 96 channels onto <= 4 (the VAE decoder's head conv) - ce_conv3d_head_bf16, see include/chronoedit_hip.h."""
    for t in list(in_frames) + list(out_frames):
        _dev(t, torch.bfloat16, "frame")
    _dev(weight, torch.bfloat16, "weight")
    assert weight.is_contiguous() and weight.numel() >= Cout * KT * 9 * Cin
    ia, oa = _ptr_array(in_frames), _ptr_array(out_frames)
    st_ev = _prof_begin()
    _check(lib().ce_conv3d_head_bf16(ia, len(in_frames), _ptr(weight), _ptr(bias), oa, len(out_frames), Cin, Cout, KT, H_out, W_out, in_Wp,
                                     out_Wp, out_border, out_cstride, out_coff, _stream()), "ce_conv3d_head_bf16")
    _prof_end(st_ev, f"conv_head_{KT}x3x3_{Cin}->{Cout}_{len(out_frames)}x{H_out}x{W_out}", 2.0 * len(out_frames) * H_out * W_out * Cout * Cin * KT * 9)


def rms_silu(x: torch.Tensor, gamma: torch.Tensor, out: torch.Tensor, T: int, C: int, H: int, W: int, in_border: int,
             out_border: int, silu: bool = True):
    _dev(x, torch.bfloat16, "x"), _dev(out, torch.bfloat16, "out"), _dev(gamma, torch.float32, "gamma")
    st = _prof_begin()
    _check(lib().ce_rms_silu_bf16(_ptr(x), _ptr(out), _ptr(gamma), T * H * W, C, H, W, in_border, out_border, int(silu), _stream()),
           "ce_rms_silu_bf16")
    _prof_end(st, f"rms_silu_{C}ch_{T}x{H}x{W}", 4.0 * T * H * W * C)  # (work: bytes read + written, bf16)
    return out


def zero_border(frames: torch.Tensor, T: int, H: int, W: int, C: int):
    """Zero the one-pixel border of T bordered channels-last frames [T, H+2, W+2, ld] (channels [0, C))."""
    _dev(frames, torch.bfloat16, "frames")
    assert frames.is_contiguous() and tuple(frames.shape[:3]) == (T, H + 2, W + 2)
    st = _prof_begin()
    _check(lib().ce_zero_border_bf16(_ptr(frames), T, H, W, C, frames.shape[3], _stream()), "ce_zero_border_bf16")
    _prof_end(st, f"zero_border_{C}ch_{T}x{H}x{W}", 2.0 * T * (2 * (W + 2) + 2 * H) * C)
    return frames


def upsample2x(x: torch.Tensor, out: torch.Tensor, T: int, C: int, H: int, W: int):
    _dev(x, torch.bfloat16, "x"), _dev(out, torch.bfloat16, "out")
    st = _prof_begin()
    _check(lib().ce_upsample2x_bf16(_ptr(x), _ptr(out), T, C, H, W, _stream()), "ce_upsample2x_bf16")
    _prof_end(st, f"upsample2x_{C}ch_{T}x{H}x{W}", 2.0 * T * H * W * C * 5)  # (read once, written four times)
    return out


def softmax_rows(scores: torch.Tensor, probs: torch.Tensor, n: int, scale: float):
    _dev(scores, torch.float32, "scores"), _dev(probs, torch.bfloat16, "probs")
    M, _, ld = _rows(scores, "scores")
    _, npad, ldp = _rows(probs, "probs")
    _check(lib().ce_softmax_rows_f32_bf16(_ptr(scores), _ptr(probs), M, n, npad, ld, ldp, float(scale), _stream()),
           "ce_softmax_rows_f32_bf16")
    return probs


_attn1_ws = {}  # (device index, floats) -> fp32 scratch of the key split of attention_1head: one per shape, allocated outside capture and NEVER
                # freed or replaced (a captured graph of an earlier shape keeps writing through its pointer)


def attention_1head(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None,
                    split_keys: bool = True):
    """softmax(q k^T * scale) v for one head of dimension C in {128, 384} (the Wan VAE mid-block): q [Nq, C], k [Nk, C] row views,
    vt [C, >= 64 ceil(Nk / 64)] = v transposed with zero padding columns; flash-style, nothing of size Nq x Nk is materialised.
    split_keys: hand the library a scratch (4 Nq (C + 2) floats per device and shape, allocated once, shared by the launches of ONE stream) so that it may split the key axis over
    several workgroups per query block when the query blocks alone do not fill the chip (ce_attention_1head_bf16)."""
    _dev(q, torch.bfloat16, "q"), _dev(k, torch.bfloat16, "k"), _dev(vt, torch.bfloat16, "vt")
    Nq, C, ldq = _rows(q, "q")
    Nk, C2, ldk = _rows(k, "k")
    Cv, _, ldvt = _rows(vt, "vt")
    assert C == C2 == Cv, (C, C2, Cv)
    if out is None:
        out = torch.empty((Nq, C), dtype=torch.bfloat16, device=q.device)
    _dev(out, torch.bfloat16, "out")
    _, _, ldo = _rows(out, "out")
    ws = None
    if split_keys:
        dev = q.device.index if q.device.index is not None else torch.cuda.current_device()
        need = 4 * Nq * (C + 2)
        ws = _attn1_ws.get((dev, need))
        if ws is None and not torch.cuda.is_current_stream_capturing():
            ws = _attn1_ws[(dev, need)] = torch.empty(need, dtype=torch.float32, device=q.device)
        # (first seen under capture: no scratch - one workgroup per query block walks all keys)
    st = _prof_begin()
    _check(lib().ce_attention_1head_bf16(_ptr(q), _ptr(k), _ptr(vt), _ptr(out), Nq, Nk, C, ldq, ldk, ldvt, ldo, float(scale), _ptr(ws),
                                         0 if ws is None else ws.numel() * 4, _stream()), "ce_attention_1head_bf16")
    _prof_end(st, f"attention_1head_{Nq}x{Nk}x{C}", 4.0 * Nq * Nk * C)
    return out


def gemm_f32(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None):
    """out[M,N] (fp32) = a[M,K] @ w[N,K]^T, no bias (CE_EPI_F32)."""
    _dev(a, torch.bfloat16, "a"), _dev(w, torch.bfloat16, "w")
    M, K, lda = _rows(a, "a")
    N, K2, ldw = _rows(w, "w")
    assert K == K2
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _, _, ldc = _rows(out, "out")
    _check(lib().ce_gemm_bf16(_ptr(a), _ptr(w), _ptr(out), _ptr(None), 4, _ptr(None), _ptr(None), M, N, K, lda, ldw, ldc, 0, 0,
                              _stream()), "ce_gemm_bf16(f32)")
    return out


# ------------------------------------------------------------------------------------------
# conditioning encoders (once per edit)
# ------------------------------------------------------------------------------------------
def gemm_batched(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, M: int, N: int, K: int, lda: int, ldw: int, ldc: int,
                 batch: tuple, stride_a: tuple, stride_w: tuple, stride_c: tuple, f32_out: bool = False,
                 bias: Optional[torch.Tensor] = None):
    """batch[0] x batch[1] products out_z[M,N] = a_z[M,K] @ w_z[N,K]^T; operand z = (z0, z1) starts stride[0]*z0 + stride[1]*z1
    elements into its tensor (flat storage offsets; see ce_gemm_batched_bf16).  Raw-pointer form: the caller states the geometry."""
    _dev(a, torch.bfloat16, "a"), _dev(w, torch.bfloat16, "w")
    _dev(out, torch.float32 if f32_out else torch.bfloat16, "out")
    if bias is not None:
        _dev(bias, torch.float32, "bias")
    st = _prof_begin()
    _check(lib().ce_gemm_batched_bf16(_ptr(a), _ptr(w), _ptr(out), _ptr(bias), EPI_F32 if f32_out else EPI_BIAS, M, N, K, lda, ldw, ldc,
                                      batch[0], batch[1], stride_a[0], stride_a[1], stride_w[0], stride_w[1], stride_c[0], stride_c[1],
                                      _stream()), "ce_gemm_batched_bf16")
    _prof_end(st, f"gemm_batched_{batch[0] * batch[1]}x{M}x{N}x{K}", 2.0 * M * N * K * batch[0] * batch[1])
    return out


def im2col_patch2d(img: torch.Tensor, patch: int, kpad: int, out: Optional[torch.Tensor] = None):
    """img [B,C,H,W] bf16 contiguous -> cols [B*(H/P)*(W/P), kpad] (column = c*P*P + y*P + x, zero padded)."""
    _dev(img, torch.bfloat16, "img")
    if img.dim() != 4 or not img.is_contiguous():
        raise ValueError("im2col_patch2d: need a contiguous [B,C,H,W] tensor")
    B, C, H, W = img.shape
    if out is None:
        out = torch.empty((B * (H // patch) * (W // patch), kpad), dtype=torch.bfloat16, device=img.device)
    _check(lib().ce_im2col_patch2d_bf16(_ptr(img), _ptr(out), B, C, H, W, patch, kpad, _stream()), "ce_im2col_patch2d_bf16")
    return out


def gather_rows(table: torch.Tensor, ids: torch.Tensor, out: Optional[torch.Tensor] = None):
    """out[i] = table[ids[i]] (bf16 rows, int64 ids)."""
    _dev(table, torch.bfloat16, "table"), _dev(ids, torch.int64, "ids")
    V, D, ldt = _rows(table, "table")
    ids = ids.reshape(-1).contiguous()
    if out is None:
        out = torch.empty((ids.numel(), D), dtype=torch.bfloat16, device=table.device)
    _, _, ldo = _rows(out, "out")
    _check(lib().ce_gather_rows_bf16(_ptr(table), _ptr(ids), _ptr(out), ids.numel(), D, ldt, ldo, V, _stream()), "ce_gather_rows_bf16")
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None):
    """T5LayerNorm: out = bf16(bf16(x * rsqrt(mean(x^2) + eps)) * w); x, w, out bf16."""
    _dev(x, torch.bfloat16, "x"), _dev(w, torch.bfloat16, "w")
    M, D, ldx = _rows(x, "x")
    assert w.numel() == D and w.is_contiguous()
    if out is None:
        out = torch.empty((M, D), dtype=torch.bfloat16, device=x.device)
    _, _, ldy = _rows(out, "out")
    st = _prof_begin()
    _check(lib().ce_rmsnorm_bf16(_ptr(x), _ptr(out), _ptr(w), M, D, ldx, ldy, float(eps), _stream()), "ce_rmsnorm_bf16")
    _prof_end(st, f"rmsnorm_{M}x{D}", 4.0 * M * D)
    return out


def softmax_t5(scores: torch.Tensor, probs: torch.Tensor, batch: int, heads: int, Lq: int, Lk: int,
               bucket_lut: Optional[torch.Tensor] = None, table: Optional[torch.Tensor] = None,
               valid_len: Optional[torch.Tensor] = None):
    """probs = softmax(scores + relative-position bias) over the valid keys; scores [batch*heads*Lq, ld] fp32, probs [.., ldp] bf16."""
    _dev(scores, torch.float32, "scores"), _dev(probs, torch.bfloat16, "probs")
    rows, _, ld = _rows(scores, "scores")
    rows_p, _, ldp = _rows(probs, "probs")
    assert rows == rows_p == batch * heads * Lq
    for t, dt, n in ((bucket_lut, torch.int32, "bucket_lut"), (valid_len, torch.int32, "valid_len"), (table, torch.float32, "table")):
        if t is not None:
            _dev(t, dt, n)
            assert t.is_contiguous()
    if table is not None:
        assert bucket_lut.numel() == Lq + Lk - 1 and table.shape[-1] == heads
    _check(lib().ce_softmax_t5_bf16(_ptr(scores), _ptr(probs), batch, heads, Lq, Lk, ld, ldp, _ptr(bucket_lut), _ptr(table),
                                    _ptr(valid_len), _stream()), "ce_softmax_t5_bf16")
    return probs


# ------------------------------------------------------------------------------------------
# fp8 (OCP e4m3) GEMM path
# ------------------------------------------------------------------------------------------
def quant_rows_fp8(x: torch.Tensor, out: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None):
    """x [M,K] bf16 -> (q [M,K] uint8 holding fp8 e4m3, s [M] fp32) with x ~= q * s[:, None] and max |q| = 448 per row."""
    _dev(x, torch.bfloat16, "x")
    M, K, ldx = _rows(x, "x")
    if out is None:
        out = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty((M,), dtype=torch.float32, device=x.device)
    _dev(out, torch.uint8, "out"), _dev(scale, torch.float32, "scale")
    _, _, ldq = _rows(out, "out")
    st = _prof_begin()
    _check(lib().ce_quant_rows_fp8(_ptr(x), _ptr(out), _ptr(scale), M, K, ldx, ldq, _stream()), "ce_quant_rows_fp8")
    _prof_end(st, f"quant_fp8_{M}x{K}", 3.0 * M * K)
    return out, scale


def ln_affine_fp8(x: torch.Tensor, a: torch.Tensor, b: torch.Tensor, eps: float, out: torch.Tensor, scale: torch.Tensor,
                  ab_rows: int = 0, ab_stride: int = 0):
    """ln_affine + quant_rows_fp8 in one pass: out (uint8 fp8 e4m3) and scale (fp32 per row) of LN(x) * a + b rounded to bf16."""
    _dev(x, torch.bfloat16, "x"), _dev(a, torch.float32, "a"), _dev(b, torch.float32, "b")
    _dev(out, torch.uint8, "out"), _dev(scale, torch.float32, "scale")
    M, D, ldx = _rows(x, "x")
    _, _, ldq = _rows(out, "out")
    st = _prof_begin()
    _check(lib().ce_ln_affine_fp8(_ptr(x), _ptr(out), _ptr(scale), _ptr(a), _ptr(b), M, D, ldx, ldq, float(eps), int(ab_rows),
                                  int(ab_stride), _stream()), "ce_ln_affine_fp8")
    _prof_end(st, f"ln_affine_fp8_{M}x{D}", 3.0 * M * D)
    return out, scale


def mx_scale_bytes(rows: int, K: int) -> int:
    """Size of the tiled E8M0 scale buffer of a [rows, K] MX operand ([ceil(rows/128)][K/128][4][16][8] bytes)."""
    return (rows + 127) // 128 * (K // 128) * 512


def mx_scales_to_rows(scale8: torch.Tensor, rows: int, K: int, w_order: bool = False) -> torch.Tensor:
    """The tiled scale buffer as a plain [rows, K/32] uint8 matrix (E8M0 bytes; for tests and tools - not on the hot path).
    w_order: the buffer is in the W order ([rb][kt][g][128 rows], `quant_rows_mxfp8(..., w_order=True)`), else in the A order."""
    rb, kt = (rows + 127) // 128, K // 128
    if w_order:
        t = scale8[: rb * kt * 512].view(rb, kt, 4, 128)            # [rb][kt][g][row % 128]
        return t.permute(0, 3, 1, 2).reshape(rb * 128, kt * 4)[:rows]
    t = scale8[: rb * kt * 512].view(rb, kt, 4, 16, 8)          # [rb][kt][g][fr][F]
    return t.permute(0, 4, 3, 1, 2).reshape(rb * 128, kt * 4)[:rows]  # row = rb*128 + F*16 + fr, block = kt*4 + g


def quant_rows_mxfp8(x: torch.Tensor, out: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None, w_order: bool = False):
    """x [M,K] bf16 -> (q [M,K] uint8 holding e4m3, scale8 uint8: one E8M0 byte per 32 consecutive elements of a row, tiled layout of
    `gemm_mxfp8`): the OCP MX contract of oracle.dit_oracle.mx_quant.  K % 128 == 0.  w_order=True: the scale order of the GEMM's WEIGHT
    operand (ce_quant_rows_mxfp8_w; weights are quantised once at load time), else of its activation operand."""
    _dev(x, torch.bfloat16, "x")
    M, K, ldx = _rows(x, "x")
    if out is None:
        out = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty((mx_scale_bytes(M, K),), dtype=torch.uint8, device=x.device)
    _dev(out, torch.uint8, "out"), _dev(scale, torch.uint8, "scale")
    assert scale.is_contiguous() and scale.numel() >= mx_scale_bytes(M, K)
    _, _, ldq = _rows(out, "out")
    st = _prof_begin()
    # (an A/B build with the staged epilogue of rounds 3-5, -DF8_EPI_LDS=1 = ce_build_info bit 2, reads the W scales in the A order)
    fn = lib().ce_quant_rows_mxfp8_w if (w_order and not lib().ce_build_info() & 4) else lib().ce_quant_rows_mxfp8
    _check(fn(_ptr(x), _ptr(out), _ptr(scale), M, K, ldx, ldq, _stream()), "ce_quant_rows_mxfp8")
    _prof_end(st, f"quant_mxfp8_{M}x{K}", 3.0 * M * K)
    return out, scale


def ln_affine_mxfp8(x: torch.Tensor, a: torch.Tensor, b: torch.Tensor, eps: float, out: torch.Tensor, scale: torch.Tensor,
                    ab_rows: int = 0, ab_stride: int = 0):
    """ln_affine + quant_rows_mxfp8 in one pass: out (uint8 e4m3) and the tiled E8M0 scales of LN(x) * a + b rounded to bf16."""
    _dev(x, torch.bfloat16, "x"), _dev(a, torch.float32, "a"), _dev(b, torch.float32, "b")
    _dev(out, torch.uint8, "out"), _dev(scale, torch.uint8, "scale")
    M, D, ldx = _rows(x, "x")
    assert scale.is_contiguous() and scale.numel() >= mx_scale_bytes(M, D)
    _, _, ldq = _rows(out, "out")
    st = _prof_begin()
    _check(lib().ce_ln_affine_mxfp8(_ptr(x), _ptr(out), _ptr(scale), _ptr(a), _ptr(b), M, D, ldx, ldq, float(eps), int(ab_rows),
                                    int(ab_stride), _stream()), "ce_ln_affine_mxfp8")
    _prof_end(st, f"ln_affine_mxfp8_{M}x{D}", 3.0 * M * D)
    return out, scale


def gemm_mxfp8(aq: torch.Tensor, sa: torch.Tensor, wq: torch.Tensor, sw: torch.Tensor, bias: Optional[torch.Tensor],
               out: Optional[torch.Tensor] = None, epilogue: int = EPI_BIAS, gate: Optional[torch.Tensor] = None,
               res: Optional[torch.Tensor] = None, gate_rows: int = 0):
    """out[M,N] (bf16) = epilogue(MX-scaled aq @ wq^T + bias); aq [M,K], wq [N,K] uint8 (e4m3) with their tiled E8M0 scale buffers: sa in the
    A order (every activation producer writes it), sw in the W order (`quant_rows_mxfp8(w, w_order=True)`)."""
    _dev(aq, torch.uint8, "aq"), _dev(wq, torch.uint8, "wq"), _dev(sa, torch.uint8, "sa"), _dev(sw, torch.uint8, "sw")
    M, K, lda = _rows(aq, "aq")
    N, K2, ldw = _rows(wq, "wq")
    if K != K2 or sa.numel() < mx_scale_bytes(M, K) or sw.numel() < mx_scale_bytes(N, K):
        raise ValueError("gemm_mxfp8: operand / scale shapes do not match")
    if bias is not None:
        _dev(bias, torch.float32, "bias")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=aq.device)
    _dev(out, torch.bfloat16, "out")
    _, _, ldc = _rows(out, "out")
    ensure_gemm_workspace(aq.device)
    ldres = 0
    if epilogue == EPI_GATE_RES:
        if res is None:
            raise ValueError("EPI_GATE_RES needs res")
        _dev(res, torch.bfloat16, "res")
        _, _, ldres = _rows(res, "res")
        if gate is not None:
            _dev(gate, torch.float32, "gate")
    st = _prof_begin()
    _check(lib().ce_gemm_mxfp8(_ptr(aq), _ptr(wq), _ptr(out), _ptr(sa), _ptr(sw), _ptr(bias), epilogue, _ptr(gate), _ptr(res), M, N, K, lda, ldw,
                               ldc, ldres, int(gate_rows), _stream()), "ce_gemm_mxfp8")
    _prof_end(st, f"gemm_mxfp8_{M}x{N}x{K}_epi{epilogue}", 2.0 * M * N * K)
    return out


def gemm_mxfp8_gelu_quant(aq: torch.Tensor, sa: torch.Tensor, wq: torch.Tensor, sw: torch.Tensor, bias: Optional[torch.Tensor],
                          out: torch.Tensor, scale: torch.Tensor):
    """quant_rows_mxfp8(gelu_tanh(aq @ wq^T + bias) rounded to bf16) with the quantisation fused into the GEMM epilogue: `out` (uint8 e4m3
    [M, N]) and `scale` (tiled E8M0 bytes of an [M, N] operand) are the next GEMM's A operand; no bf16 matrix is written.  N % 128 == 0."""
    _dev(aq, torch.uint8, "aq"), _dev(wq, torch.uint8, "wq"), _dev(sa, torch.uint8, "sa"), _dev(sw, torch.uint8, "sw")
    _dev(out, torch.uint8, "out"), _dev(scale, torch.uint8, "scale")
    M, K, lda = _rows(aq, "aq")
    N, K2, ldw = _rows(wq, "wq")
    Mo, No, ldq = _rows(out, "out")
    if K != K2 or (Mo, No) != (M, N) or sa.numel() < mx_scale_bytes(M, K) or sw.numel() < mx_scale_bytes(N, K) or scale.numel() < mx_scale_bytes(M, N):
        raise ValueError("gemm_mxfp8_gelu_quant: operand / scale shapes do not match")
    if bias is not None:
        _dev(bias, torch.float32, "bias")
    ensure_gemm_workspace(aq.device)  # (its launcher cuts the last round along K like the others: the stream's own scratch must be registered)
    st = _prof_begin()
    _check(lib().ce_gemm_mxfp8_gelu_quant(_ptr(aq), _ptr(wq), _ptr(sa), _ptr(sw), _ptr(bias), _ptr(out), _ptr(scale), M, N, K, lda, ldw, ldq,
                                          _stream()), "ce_gemm_mxfp8_gelu_quant")
    _prof_end(st, f"gemm_mxfp8_{M}x{N}x{K}_gelu_quant", 2.0 * M * N * K)
    return out, scale


def set_gemm_fp8_variant(v: int) -> int:
    """Main loop of `gemm_fp8`: 0 = 8 waves / 4 phases, 1 = one wave per SIMD (ce_gemm_fp8w4.hip); returns the previous setting."""
    return _set_knob("ce_set_gemm_fp8_variant", v)


def gemm_fp8(aq: torch.Tensor, sa: torch.Tensor, wq: torch.Tensor, sw: torch.Tensor, bias: Optional[torch.Tensor],
             out: Optional[torch.Tensor] = None, epilogue: int = EPI_BIAS, gate: Optional[torch.Tensor] = None,
             res: Optional[torch.Tensor] = None, gate_rows: int = 0):
    """out[M,N] (bf16) = epilogue(sa[:,None] * sw[None,:] * (aq @ wq^T) + bias); aq [M,K], wq [N,K] uint8 (fp8 e4m3)."""
    _dev(aq, torch.uint8, "aq"), _dev(wq, torch.uint8, "wq"), _dev(sa, torch.float32, "sa"), _dev(sw, torch.float32, "sw")
    M, K, lda = _rows(aq, "aq")
    N, K2, ldw = _rows(wq, "wq")
    if K != K2 or sa.numel() != M or sw.numel() != N:
        raise ValueError("gemm_fp8: operand / scale shapes do not match")
    if bias is not None:
        _dev(bias, torch.float32, "bias")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=aq.device)
    _dev(out, torch.bfloat16, "out")
    _, _, ldc = _rows(out, "out")
    ensure_gemm_workspace(aq.device)  # the one-wave-per-SIMD loop cuts a partially filled last round of tiles along K (as ce_gemm_bf16)
    ldres = 0
    if epilogue == EPI_GATE_RES:
        if res is None:
            raise ValueError("EPI_GATE_RES needs res")
        _dev(res, torch.bfloat16, "res")
        _, _, ldres = _rows(res, "res")
        if gate is not None:
            _dev(gate, torch.float32, "gate")
    st = _prof_begin()
    _check(lib().ce_gemm_fp8(_ptr(aq), _ptr(wq), _ptr(out), _ptr(sa), _ptr(sw), _ptr(bias), epilogue, _ptr(gate), _ptr(res), M, N, K,
                             lda, ldw, ldc, ldres, int(gate_rows), _stream()), "ce_gemm_fp8")
    _prof_end(st, f"gemm_fp8_{M}x{N}x{K}_epi{epilogue}", 2.0 * M * N * K)
    return out


# ------------------------------------------------------------------------------------------
# MXFP8 self-attention (fp8 mode; contract: chronoedit_amd/csrc/ce_attn_fp8.hip, oracle.dit_oracle.attention_mxfp8)
# ------------------------------------------------------------------------------------------
MXFP8_Q_SCALE = 128 ** -0.5 * 1.4426950408889634  # softmax_scale * log2(e): what the q producer multiplies in (head_dim 128)


def rmsnorm_rope_mxfp8(x: torch.Tensor, w: torch.Tensor, cos_sin: Optional[torch.Tensor], head_dim: int, eps: float,
                       out: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None, post_scale: float = 1.0):
    """RMSNorm across heads (+ RoPE) of x [M, D] (bf16, row stride free), times post_scale -> (q8 [M, D] uint8 e4m3, s8 [M, D/32]
    uint8 E8M0): MXFP8 blocks of 32 head channels.  The attention kernel expects q produced with post_scale = MXFP8_Q_SCALE
    (scores then come out of the matrix pipe in the exp2 domain) and k with 1."""
    _dev(x, torch.bfloat16, "x"), _dev(w, torch.float32, "w")
    M, D, ldx = _rows(x, "x")
    rope_rows = 0
    if cos_sin is not None:
        _dev(cos_sin, torch.float32, "cos_sin")
        assert cos_sin.is_contiguous() and cos_sin.shape[1:] == (head_dim // 2, 2) and M % cos_sin.shape[0] == 0
        rope_rows = cos_sin.shape[0]
    if out is None:
        out = torch.empty((M, D), dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty((M, D // 32), dtype=torch.uint8, device=x.device)
    _dev(out, torch.uint8, "out"), _dev(scale, torch.uint8, "scale")
    assert scale.is_contiguous() and scale.shape == (M, D // 32)
    _, _, ldq = _rows(out, "out")
    st = _prof_begin()
    _check(lib().ce_rmsnorm_rope_mxfp8(_ptr(x), _ptr(w), _ptr(cos_sin), _ptr(out), _ptr(scale), M, D, ldx, ldq, head_dim, float(eps), rope_rows,
                                       float(post_scale), _stream()), "ce_rmsnorm_rope_mxfp8")
    _prof_end(st, f"rmsnorm_rope_mxfp8_{M}x{D}", 3.0 * M * D)
    return out, scale


def v_mxfp8_transpose(v: torch.Tensor, n_tokens: int, batch: int, heads: int, out: Optional[torch.Tensor] = None,
                      scale: Optional[torch.Tensor] = None):
    """v [batch*n_tokens, heads*128] bf16 (row stride free) -> (v8t [batch, heads, 128, npad] uint8, sv [batch, heads, npad/64, 128, 2] uint8),
    npad = n_tokens rounded up to 64: V^T tiles in the key order of the attention kernel's P operand, MXFP8 blocks of 32 consecutive keys
    (sv[.., t, d, beta] = E8M0 scale of keys 64 t + 32 beta .. of channel d)."""
    _dev(v, torch.bfloat16, "v")
    Mv, Dv, ldv = _rows(v, "v")
    assert Mv == batch * n_tokens and Dv == heads * 128
    npad = (n_tokens + 63) // 64 * 64
    if out is None:
        out = torch.empty((batch, heads, 128, npad), dtype=torch.uint8, device=v.device)
    if scale is None:
        scale = torch.empty((batch, heads, npad // 64, 128, 2), dtype=torch.uint8, device=v.device)
    assert out.is_contiguous() and scale.is_contiguous() and out.shape == (batch, heads, 128, npad) and scale.shape == (batch, heads, npad // 64, 128, 2)
    st = _prof_begin()
    _check(lib().ce_v_mxfp8_transpose(_ptr(v), ldv, _ptr(out), _ptr(scale), n_tokens, batch, heads, npad, _stream()), "ce_v_mxfp8_transpose")
    _prof_end(st, f"v_mxfp8_transpose_{Mv}x{Dv}", 3.0 * Mv * Dv)
    return out, scale


def set_attention_mxfp8_variant(v: int) -> int:
    """0: plain loop (exact running maximum every tile), 1: software-pipelined with the speculative offset (default); returns the
    previous setting.  (A one-wave-per-SIMD form, 4 waves x 64 rows, measured 1.20 vs 1.71 PFLOP/s at 28 800 keys and was removed:
    profiles/r02_microbench_attn_mxfp8.txt.)"""
    return _set_knob("ce_set_attention_mxfp8_variant", v)


def set_attention_mxfp8_persistent(n: int) -> int:
    """(diagnostic build) workgroups of the persistent MXFP8 attention form: 0 = one workgroup per work item, default 512; returns the previous value."""
    return _set_knob("ce_set_attention_mxfp8_persistent", n)


def attention_mxfp8(q8: torch.Tensor, sq: torch.Tensor, k8: torch.Tensor, sk: torch.Tensor, v8t: torch.Tensor, sv: torch.Tensor,
                    heads: int, out: Optional[torch.Tensor] = None, batch: int = 1, out8: Optional[torch.Tensor] = None,
                    scale8: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None):
    """Attention on the MX-fp8 matrix instruction from the operands the two producers above write (q with post_scale =
    MXFP8_Q_SCALE); out [batch*Nq, heads*128] bf16 - or, with out8 / scale8, the same values as the MX fp8 operand of the out-projection
    (e4m3 rows + tiled E8M0 block scales, bit-identical to quant_rows_mxfp8 of the bf16 output).
    add: bf16 rows [batch*Nq, heads*128] added to this call's bf16-rounded result before it is stored / quantised - the second segment of the
    cross-attention (`add` = the text segment's result; `out` may be `add` itself)."""
    for n, t in (("q8", q8), ("sq", sq), ("k8", k8), ("sk", sk), ("v8t", v8t), ("sv", sv)):
        _dev(t, torch.uint8, n)
    Mq, D, ldq = _rows(q8, "q8")
    Mk, _, ldk = _rows(k8, "k8")
    assert D == heads * 128 and Mq % batch == 0 and Mk % batch == 0
    nq, nkv = Mq // batch, Mk // batch
    npad = v8t.shape[-1]
    assert sq.is_contiguous() and sk.is_contiguous() and v8t.is_contiguous() and sv.is_contiguous()
    assert sq.shape == (Mq, D // 32) and sk.shape == (Mk, D // 32) and v8t.shape == (batch, heads, 128, npad) and sv.shape == (batch, heads, npad // 64, 128, 2)
    if add is not None:
        _dev(add, torch.bfloat16, "add")
        Ma, Da, lda = _rows(add, "add")
        assert (Ma, Da) == (Mq, D)
        ld8 = ldo = 0
        if out8 is not None:
            _dev(out8, torch.uint8, "out8"), _dev(scale8, torch.uint8, "scale8")
            _, D8, ld8 = _rows(out8, "out8")
            assert out8.shape[0] == Mq and D8 == D and scale8.is_contiguous() and scale8.numel() >= mx_scale_bytes(Mq, D)
        else:
            if out is None:
                out = add
            _dev(out, torch.bfloat16, "out")
            _, _, ldo = _rows(out, "out")
        st = _prof_begin()
        _check(lib().ce_attention_mxfp8_add(_ptr(q8), _ptr(sq), _ptr(k8), _ptr(sk), _ptr(v8t), _ptr(sv), _ptr(add), lda, _ptr(out) if out8 is None else None,
                                            ldo, _ptr(out8), _ptr(scale8) if out8 is not None else None, ld8, nq, nkv, npad, heads, 128, ldq, ldk, batch,
                                            _stream()), "ce_attention_mxfp8_add")
        _prof_end(st, f"attention_mxfp8_{nq}x{nkv}_h{heads}" + (f"_b{batch}" if batch > 1 else "") + "_add" + ("_mxq" if out8 is not None else ""),
                  4.0 * nq * nkv * 128 * heads * batch)
        return out8 if out8 is not None else out
    if out8 is not None:
        _dev(out8, torch.uint8, "out8"), _dev(scale8, torch.uint8, "scale8")
        _, D8, ld8 = _rows(out8, "out8")
        assert out8.shape[0] == Mq and D8 == D and scale8.is_contiguous() and scale8.numel() >= mx_scale_bytes(Mq, D)
        st = _prof_begin()
        _check(lib().ce_attention_mxfp8_quant(_ptr(q8), _ptr(sq), _ptr(k8), _ptr(sk), _ptr(v8t), _ptr(sv), _ptr(out8), _ptr(scale8), nq, nkv, npad,
                                              heads, 128, ldq, ldk, ld8, batch, _stream()), "ce_attention_mxfp8_quant")
        _prof_end(st, f"attention_mxfp8_{nq}x{nkv}_h{heads}" + (f"_b{batch}" if batch > 1 else "") + "_mxq", 4.0 * nq * nkv * 128 * heads * batch)
        return out8
    if out is None:
        out = torch.empty((Mq, D), dtype=torch.bfloat16, device=q8.device)
    _dev(out, torch.bfloat16, "out")
    _, _, ldo = _rows(out, "out")
    st = _prof_begin()
    _check(lib().ce_attention_mxfp8(_ptr(q8), _ptr(sq), _ptr(k8), _ptr(sk), _ptr(v8t), _ptr(sv), _ptr(out), nq, nkv, npad, heads, 128, ldq, ldk, ldo,
                                    batch, _stream()), "ce_attention_mxfp8")
    _prof_end(st, f"attention_mxfp8_{nq}x{nkv}_h{heads}" + (f"_b{batch}" if batch > 1 else ""), 4.0 * nq * nkv * 128 * heads * batch)
    return out
