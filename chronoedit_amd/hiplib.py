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
LIB_PATH = os.path.join(LIB_DIR, "libchronoedit_hip.so")            # the product library: exports exactly include/chronoedit_hip.h
DIAG_LIB_PATH = os.path.join(LIB_DIR, "libchronoedit_hip_diag.so")  # same sources + -DCE_DIAGNOSTICS: also exports chronoedit_hip_diag.h
OBJ_DIR = os.path.join(LIB_DIR, "obj")
HEADER = os.path.join(ROOT, "include", "chronoedit_hip.h")
DIAG_HEADER = os.path.join(ROOT, "include", "chronoedit_hip_diag.h")

# translation units that hold a kernel-body selector (csrc/ce_common.h CE_KNOB): the only ones compiled twice
DIAG_SOURCES = ["ce_gemm.hip", "ce_gemm256.hip", "ce_attn.hip", "ce_attn_fp8.hip", "ce_gemm_fp8.hip", "ce_gemm_fp8w4.hip"]
# measured-and-closed experiments kept as opt-in bodies: in the diagnostic library only
DIAG_ONLY_SOURCES = ["ce_attn16.hip"]
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
    "ce_rope_scatter_bf16": [_P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _I, _F, _I, _P],
    "ce_patchify_rows_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
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
    "ce_quant_rows_mxfp8": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ce_quant_rows_mxfp8_w": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ce_ln_affine_mxfp8": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P],
    "ce_gemm_mxfp8": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ce_gemm_mxfp8_gelu_quant": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ce_gemm_batched_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I] + [ctypes.c_longlong] * 6 + [_P],
    "ce_im2col_patch2d_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ce_gather_rows_bf16": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ce_rmsnorm_bf16": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    "ce_softmax_t5_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
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
    "ce_conv3d_gemm_rms_silu_bf16": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "ce_rms_silu_bf16": [_P, _P, _P, _c.c_longlong, _I, _I, _I, _I, _I, _I, _P],
    "ce_upsample2x_bf16": [_P, _P, _I, _I, _I, _I, _P],
    "ce_softmax_rows_f32_bf16": [_P, _P, _I, _I, _I, _I, _I, _F, _P],
    "ce_attention_1head_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _c.c_longlong, _P],
    "ce_cfg_unipc_step": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_longlong, _I, _P],
    "ce_build_info": [],
}
# the selectors only libchronoedit_hip_diag.so exports (include/chronoedit_hip_diag.h)
DIAG_SIGNATURES: Dict[str, List] = {
    "ce_set_gemm_variant": [_I],
    "ce_set_attention_waves": [_I],
    "ce_set_gemm_fp8_variant": [_I],
    "ce_set_attention_mxfp8_variant": [_I],
    "ce_set_attention_mxfp8_persistent": [_I],
    "ce_diag_attention_exact_route_hits": [_P, _I],
}


def header_symbols(diag: bool = False) -> List[str]:
    """Every function declared in include/chronoedit_hip.h (diag: in include/chronoedit_hip_diag.h)."""
    txt = open(DIAG_HEADER if diag else HEADER).read()
    return re.findall(r"^int (ce_\w+)\(", txt, flags=re.M)


def exported_symbols(path: str) -> List[str]:
    """The dynamic symbols a built library defines (`nm -D --defined-only`): tests compare them with the headers."""
    nm = "/opt/rocm/lib/llvm/bin/llvm-nm" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-nm") else "nm"
    out = subprocess.run([nm, "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-2] in ("T", "t", "W", "D", "B"))


HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"]


def build(force: bool = False, verbose: bool = False, jobs: int = 0) -> str:
    """hipcc --offload-arch=gfx950 every csrc/*.hip into lib/libchronoedit_hip.so and - the translation units that hold a kernel-body
    selector compiled a second time with -DCE_DIAGNOSTICS - lib/libchronoedit_hip_diag.so (cross-compiles without a GPU).  One object
    per translation unit under lib/obj/, compiled in parallel, rebuilt when its source or any header is newer.  Always writes the in-tree
    paths (CE_HIPLIB_PATH only redirects load())."""
    import glob
    from concurrent.futures import ThreadPoolExecutor
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [HEADER, DIAG_HEADER]
    hdr_m = max(os.path.getmtime(d) for d in hdrs)
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    todo, objs, diag_objs = [], [], []
    for s_ in srcs:
        src, stem = os.path.join(CSRC, s_), s_[:-4]
        variants = [("", [])] + ([(".diag", ["-DCE_DIAGNOSTICS"])] if s_ in DIAG_SOURCES else [])
        for tag, defs in variants:
            obj = os.path.join(OBJ_DIR, stem + tag + ".o")
            if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
                todo.append([hipcc, *HIPCC_FLAGS, *defs, "-I", CSRC, "-c", src, "-o", obj])
            if tag or s_ in DIAG_ONLY_SOURCES:
                diag_objs.append(obj)
            else:
                objs.append(obj)
        if s_ not in DIAG_SOURCES and s_ not in DIAG_ONLY_SOURCES:
            diag_objs.append(os.path.join(OBJ_DIR, stem + ".o"))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 4)) as ex:
            list(ex.map(run, todo))
    # the dynamic symbol table is the header, nothing else: a linker version script generated FROM the header(s) keeps the toolchain's own
    # globals (__hip_cuid_*, kernel host stubs) out of it (tests/test_host_cpu.py compares `nm -D` with the headers)
    for path, group, diag in ((LIB_PATH, objs, False), (DIAG_LIB_PATH, diag_objs, True)):
        if force or todo or not os.path.exists(path) or os.path.getmtime(path) < max([hdr_m] + [os.path.getmtime(o) for o in group]):
            vs = os.path.join(OBJ_DIR, "exports_diag.map" if diag else "exports.map")
            names = header_symbols() + (header_symbols(diag=True) if diag else [])
            with open(vs, "w") as f:
                f.write("{\n  global:\n" + "".join(f"    {n};\n" for n in names) + "  local: *;\n};\n")
            run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *group, "-ldl", f"-Wl,--version-script={vs}", "-o", path])
    return LIB_PATH


_LIB = None
_DIAG = None


def _open(path: str, diag: bool) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension is the only compute path of chronoedit_amd "
            "(no CPU/eager fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'`."
        )
    # ONE HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 and must be the first to load it - our library's
    # DT_NEEDED then resolves to that copy.  Loaded the other way round (this library before `import torch`) the process holds
    # two runtimes, and a kernel launched through one on a stream of the other fails with hipErrorNoDevice (100).
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    sigs = dict(SIGNATURES, **(DIAG_SIGNATURES if diag else {}))
    for name in header_symbols() + (header_symbols(diag=True) if diag else []):
        if not hasattr(lib, name):
            raise RuntimeError(f"{path} does not export {name} declared in include/chronoedit_hip{'_diag' if diag else ''}.h")
    for name, argtypes in sigs.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError(f"{path} does not export {name}")
        fn.argtypes = argtypes
        fn.restype = _c.c_int
    return lib


def load() -> ctypes.CDLL:
    """The product library.  CE_HIPLIB_PATH (A/B tooling: tools/sessions_r05/gpu_r5_l.sh runs the same bench against two BUILDS) redirects the
    load - with a warning - and never the build; a build whose ce_build_info() says F8_ABLATE != 0 (results are garbage) or that is the
    diagnostic build is refused unless CE_HIPLIB_ALLOW_DIAGNOSTIC_BUILD=1 says the caller knows."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = LIB_PATH
    override = os.environ.get("CE_HIPLIB_PATH")
    if override:
        import warnings
        warnings.warn(f"CE_HIPLIB_PATH is set: loading {override} instead of the in-tree build {LIB_PATH}", RuntimeWarning, stacklevel=2)
        path = override
    lib = _open(path, diag=False)
    info = lib.ce_build_info()
    if (info & 1 or (info >> 8) & 255) and os.environ.get("CE_HIPLIB_ALLOW_DIAGNOSTIC_BUILD") != "1":
        raise RuntimeError(f"{path}: ce_build_info() = {info:#x} - " + ("a timing-only F8_ABLATE build whose results are garbage" if (info >> 8) & 255
                           else "the diagnostic build") + "; refusing to use it as the product library (CE_HIPLIB_ALLOW_DIAGNOSTIC_BUILD=1 overrides)")
    _LIB = lib
    return lib


def load_diagnostics() -> ctypes.CDLL:
    """libchronoedit_hip_diag.so: the product ABI plus the kernel-body selectors (tools/, body-equivalence tests; never the product path)."""
    global _DIAG
    if _DIAG is None:
        _DIAG = _open(DIAG_LIB_PATH, diag=True)
        if not _DIAG.ce_build_info() & 1:
            raise RuntimeError(f"{DIAG_LIB_PATH} was not compiled with -DCE_DIAGNOSTICS")
    return _DIAG
