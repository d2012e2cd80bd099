"""Build + load libchronoedit_hip.so (the C-ABI HIP library) through ctypes.

The library is built in-tree (``chronoedit_amd/lib/``) with ``hipcc --offload-arch=gfx950`` so that
it travels with the source tree; nothing is JIT-compiled at run time.  Loading fails loudly when
the library is missing or lacks a symbol declared in include/chronoedit_hip.h.
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess
from typing import Dict, List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
# CE_HIPLIB_PATH: A/B tooling only (tools/sessions_r05/gpu_r5_l.sh runs the same bench against two BUILDS of the library); the product loads the in-tree build
LIB_PATH = os.environ.get("CE_HIPLIB_PATH") or os.path.join(LIB_DIR, "libchronoedit_hip.so")
HEADER = os.path.join(ROOT, "include", "chronoedit_hip.h")

SOURCES = ["ce_rowops.hip", "ce_gemm.hip", "ce_gemm256.hip", "ce_gemm256w4.hip", "ce_gemm384.hip", "ce_attn.hip", "ce_attn16.hip", "ce_attn_fp8.hip", "ce_sched.hip", "ce_conv.hip", "ce_enc.hip", "ce_gemm_fp8.hip", "ce_gemm_fp8w4.hip", "ce_comm.hip"]

_c = ctypes
_P, _I, _F = _c.c_void_p, _c.c_int, _c.c_float

# symbol -> argtypes (restype is always int)
SIGNATURES: Dict[str, List] = {
    "ce_ln_affine_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P],
    "ce_rmsnorm_rope_bf16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "ce_gemm_bf16": [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ce_gemm_aseg_bf16": [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _c.c_longlong, _P],
    "ce_gemm_seg_bf16": [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _c.c_longlong, _I, _c.c_longlong, _P],
    "ce_rmsnorm_rope_mxfp8": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _F, _P],
    "ce_v_mxfp8_transpose": [_P, _I, _P, _P, _I, _I, _I, _I, _P],
    "ce_attention_mxfp8": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ce_attention_mxfp8_quant": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ce_attention_mxfp8_add": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ce_set_attention_mxfp8_variant": [_I],
    "ce_set_attention_mxfp8_persistent": [_I],
    "ce_rope_scatter_bf16": [_P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _I, _F, _I, _P],
    "ce_patchify_rows_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ce_set_gemm_variant": [_I],
    "ce_set_gemm_workspace": [_P, ctypes.c_size_t],
    "ce_set_gemm_workspace_stream": [_P, _P, ctypes.c_size_t],
    "ce_comm_load": [ctypes.c_char_p],
    "ce_comm_unique_id": [_P],
    "ce_comm_init": [ctypes.POINTER(ctypes.c_void_p), _P, _I, _I],
    "ce_comm_destroy": [_P],
    "ce_comm_all_to_all": [_P, _P, _P, ctypes.c_size_t, _P],
    "ce_comm_all_gather": [_P, _P, _P, ctypes.c_size_t, _P],
    "ce_ln_affine_fp8": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P],
    "ce_quant_rows_fp8": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ce_gemm_fp8": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ce_set_gemm_fp8_variant": [_I],
    "ce_quant_rows_mxfp8": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ce_ln_affine_mxfp8": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P],
    "ce_gemm_mxfp8": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ce_gemm_mxfp8_gelu_quant": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ce_gemm_batched_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I] + [ctypes.c_longlong] * 6 + [_P],
    "ce_im2col_patch2d_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ce_gather_rows_bf16": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ce_rmsnorm_bf16": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    "ce_softmax_t5_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "ce_set_attention_waves": [_I],
    "ce_attention_bf16": [_P, _P, _P, _I, _I, _I, _P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P],
    "ce_attention_batched_bf16": [_P, _P, _P, _I, _I, _I, _P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "ce_attention_vt_bf16": [_P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "ce_attention_2seg_vt_bf16": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "ce_attention_2seg_vt_quant_bf16": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "ce_v_transpose_bf16": [_P, _I, _P, _I, _I, _I, _P],
    "ce_attention_vt_blocked_bf16": [_P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _F, _I, _I, _I, _I, _P],
    "ce_v_transpose_blocked_bf16": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ce_timestep_sinusoid": [_P, _P, _I, _P],
    "ce_timestep_sinusoid_f32": [_P, _P, _I, _P],
    "ce_gemv": [_P, _I, _P, _P, _P, _I, _I, _I, _P],
    "ce_modulation": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ce_patchify_bf16": [_P, _P, _I, _I, _I, _I, _I, _P],
    "ce_unpatchify_bf16": [_P, _P, _I, _I, _I, _I, _I, _P],
    "ce_conv_igemm_bf16": [_P, _I, _P, _P, _P, _I, _P] + [_I] * 16 + [_P],
    "ce_conv3d_head_bf16": [_P, _I, _P, _P, _P, _I] + [_I] * 10 + [_P],
    "ce_gemm_bf16_tile_rows": [_I, _I, _I, _I, _c.c_longlong],
    "ce_zero_border_bf16": [_P, _I, _I, _I, _I, _I, _P],
    "ce_conv3d_gemm_bf16": [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ce_rms_silu_bf16": [_P, _P, _P, _c.c_longlong, _I, _I, _I, _I, _I, _I, _P],
    "ce_upsample2x_bf16": [_P, _P, _I, _I, _I, _I, _P],
    "ce_softmax_rows_f32_bf16": [_P, _P, _I, _I, _I, _I, _I, _F, _P],
    "ce_attention_1head_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "ce_cfg_unipc_step": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_longlong, _I, _P],
}


def header_symbols() -> List[str]:
    """Every function declared in include/chronoedit_hip.h."""
    txt = open(HEADER).read()
    return re.findall(r"^int (ce_\w+)\(", txt, flags=re.M)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 every csrc/*.hip into lib/libchronoedit_hip.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    import glob
    deps = srcs + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [HEADER]  # every header a source may include
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
            return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CSRC, *srcs, "-ldl", "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


_LIB = None


def load() -> ctypes.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is the only compute path of chronoedit_amd "
            "(no CPU/eager fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'`."
        )
    # ONE HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 and must be the first to load it - our library's
    # DT_NEEDED then resolves to that copy.  Loaded the other way round (this library before `import torch`) the process holds
    # two runtimes, and a kernel launched through one on a stream of the other fails with hipErrorNoDevice (100).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name in header_symbols():
        if not hasattr(lib, name):
            raise RuntimeError(f"{LIB_PATH} does not export {name} declared in include/chronoedit_hip.h")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError(f"{LIB_PATH} does not export {name}")
        fn.argtypes = argtypes
        fn.restype = _c.c_int
    _LIB = lib
    return lib
