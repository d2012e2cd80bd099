"""ISA lint for the HIP sources (no GPU needed): compiles every csrc/*.hip to gfx950 assembly and reports, per kernel,
  * registers / scratch (a spill in a hot kernel is a bug; 2 waves per SIMD need <= 256 VGPRs, 3 need <= 168),
  * "serialised loads": a global / buffer load whose very next memory-counter wait is `s_waitcnt vmcnt(0)` within a few
    instructions - what hipcc emits for a load inside a per-element bounds guard (one load in flight per wave; this cost the
    row kernels 15-25 % and the gated-residual GEMM epilogue 5 %),
  * MFMA count, s_nop count and v_mov count (copies the register allocator inserted).
    python tools/isa_lint.py [file.hip ...] [--min-insts N]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chronoedit_amd", "csrc")


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
            if len(out) >= len(names):
                return {n: re.sub(r"\(anonymous namespace\)::", "", o) for n, o in zip(names, out)}
        except OSError:
            continue
    return {n: n for n in names}


def loop_scratch_free(path, kernel_substr):
    """True when no basic block of the kernel that sits in a loop (`Depth=`) and issues MFMAs touches scratch."""
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-S", "--cuda-device-only", path, "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    m = re.search(r"^([A-Za-z0-9_]*" + re.escape(kernel_substr) + r"[A-Za-z0-9_]*):", text, flags=re.M)
    body = text[m.end():text.index(".Lfunc_end", m.end())]
    parts = re.split(r"\n(\.LBB[0-9_]+:|; %bb\.[0-9]+:[^\n]*)", body)
    for blk in parts[2::2]:
        if "v_mfma" in blk and "Depth=2" in blk and "scratch_" in blk:
            return False
    return True


def inner_loops(path, kernel_substr):
    """[(kernel, {mnemonic: count})] for the first innermost loop (its header block plus every block the assembler annotates as
    `in Loop: Header=<that block>` - hipcc rotates loops, so part of the body may sit in front of the header) of every kernel whose
    mangled name contains `kernel_substr`: what the K loop of a GEMM actually issues per trip."""
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-S", "--cuda-device-only", path, "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    out = []
    for m in re.finditer(r"^(\S*" + re.escape(kernel_substr) + r"\S*):\s*; @\1$", text, flags=re.M):
        body = text[m.end():text.index(".Lfunc_end", m.end())]
        h = re.search(r"^\.L(BB[0-9_]+):\s*; =>This Inner Loop Header", body, flags=re.M)
        if not h:
            continue
        hdr = h.group(1)
        counts, inside = {}, False
        for l in body.split("\n"):
            s = l.strip()
            if re.match(r"^(\.LBB[0-9_]+:|; %bb\.[0-9]+:)", s):  # a block starts: does it belong to the loop?
                inside = s.startswith(".L" + hdr + ":") or ("in Loop: Header=" + hdr + " ") in s
                continue
            if inside and s and s[0] not in ";." and not s.endswith(":"):
                op = s.split()[0]
                counts[op] = counts.get(op, 0) + 1
        out.append((m.group(1), counts))
    return out


def lint(path, min_insts):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-S", "--cuda-device-only", path, "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    meta = {}
    for m in re.finditer(r"\.set (\S+)\.(num_vgpr|num_agpr|private_seg_size), (\d+)", text):
        meta.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
    kernels = re.findall(r"^(\S+):\s*; @\1$", text, flags=re.M)
    pretty = demangle(kernels)
    rows = []
    for k in kernels:
        body = text[text.index(f"\n{k}:"):]
        end = body.find(".Lfunc_end")  # (not the first s_endpgm: block placement can put an exit path in the middle of a function)
        body = body[:end] if end >= 0 else body[:body.index("s_endpgm")]
        lines = [l.strip() for l in body.split("\n") if l.startswith("\t")]
        lines = [l for l in lines if l and not l.startswith((";", "."))]
        ops = [l.split()[0] for l in lines]
        if len(ops) < min_insts:
            continue
        serial = 0
        for i, op in enumerate(ops):
            if not op.startswith(("global_load", "buffer_load")) or " lds" in lines[i]:
                continue
            for l in lines[i + 1:i + 6]:  # the first wait behind the load, unless another load comes first
                if l.startswith(("global_load", "buffer_load")):
                    break
                if l.startswith("s_waitcnt"):
                    serial += "vmcnt(0)" in l
                    break
        mm = meta.get(k, {})
        rows.append((pretty[k][:78], len(ops), mm.get("num_vgpr", 0) + mm.get("num_agpr", 0), mm.get("private_seg_size", 0),
                     sum(op.startswith("v_mfma") for op in ops), serial, sum(op == "s_nop" for op in ops),
                     sum(op.startswith("v_mov_b") for op in ops)))
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    min_insts = int(sys.argv[sys.argv.index("--min-insts") + 1]) if "--min-insts" in sys.argv else 60
    files = args or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    print(f"{'kernel':78s} {'insts':>6s} {'regs':>5s} {'scratch':>7s} {'mfma':>5s} {'ser.ld':>6s} {'s_nop':>5s} {'v_mov':>5s}")
    for f in files:
        print(f"-- {os.path.relpath(f, ROOT)}")
        for r in lint(f, min_insts):
            flag = "  <-- spill" if r[3] else ("  <-- serialised loads" if r[5] >= 4 else "")
            print(f"{r[0]:78s} {r[1]:6d} {r[2]:5d} {r[3]:7d} {r[4]:5d} {r[5]:6d} {r[6]:5d} {r[7]:5d}{flag}")


if __name__ == "__main__":
    main()
