"""Build a VARIANT of libchronoedit_hip.so for the A/B tools: one or more translation units recompiled with extra -D flags, linked with the product
objects of every other unit (chronoedit_amd/lib/obj/, built by hiplib.build()) into chronoedit_amd/lib/lib<tag>.so.
    python tools/build_variant.py <tag> <unit.hip>[@<other source>][,<unit.hip>...] [-DNAME=VALUE ...]
e.g. python tools/build_variant.py ce_f8epilds ce_gemm_fp8w4.hip -DF8_EPI_LDS=1      (the staged epilogue of rounds 3-5)
     git show <rev>:chronoedit_amd/csrc/ce_gemm384.hip > /tmp/old384.hip; python tools/build_variant.py ce_g384_r5 ce_gemm384.hip@/tmp/old384.hip"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import hiplib  # noqa: E402


def main():
    tag = sys.argv[1]
    units = dict((u.split("@") + [None])[:2] for u in sys.argv[2].split(","))  # unit -> alternative source file (None: the tree's)
    defs = sys.argv[3:]
    hiplib.build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in hiplib.SOURCES:
        if s in hiplib.DIAG_ONLY_SOURCES:
            continue
        if s in units:
            obj = os.path.join(hiplib.OBJ_DIR, f"{s[:-4]}.{tag}.o")
            subprocess.run([hipcc, *hiplib.HIPCC_FLAGS, *defs, "-I", hiplib.CSRC, "-x", "hip", "-c", units[s] or os.path.join(hiplib.CSRC, s), "-o", obj], check=True)
        else:
            obj = os.path.join(hiplib.OBJ_DIR, s[:-4] + ".o")
        objs.append(obj)
    out = os.path.join(hiplib.LIB_DIR, f"lib{tag}.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", f"-Wl,--version-script={os.path.join(hiplib.OBJ_DIR, 'exports.map')}",
                    "-o", out], check=True)
    print(out)


if __name__ == "__main__":
    main()
