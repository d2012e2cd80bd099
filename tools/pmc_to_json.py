"""profiles/r06_pmc_*.txt / r05_pmc_*.txt (the summaries the GPU sessions write: one counter set per rocprofv3 pass) -> profiles/r06_pmc_traffic.json, the
file bench.py::_pmc_traffic reads `roofline.traffic` from.  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes (MI355X_MICROARCH.md, HBM section): fetch_bytes = FETCH_SIZE x 1024 x 2.  Effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA busy =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).    python tools/pmc_to_json.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")

# file, kernel-name fragment, bench labels that map to it, algorithmic bytes per launch, description
SPECS = [
    ("r05_pmc_attention_vt_7200_b2.txt", "attn_fwd_sp_kernel", ["attention_7200x7200+0_h40_b2"], 589824000,
     "attn_fwd_sp_kernel<false, VT = true> (ce_attention_vt_bf16: K and V^T by LDS-DMA), 7200 keys x 40 heads x 2 samples", "tools/one_kernel.py attnvt 7200 40 2"),
    ("r06_pmc_gemm_outproj.txt", "gemm_bf16_384ILi2E", ["gemm_14400x5120x5120_epi2"], 494796800,
     "gemm_bf16_384<EPI_GATE_RES> (384 x 256 macro tile, one wave per SIMD) 14400 x 5120 x 5120", "tools/one_kernel.py gemm 14400 5120 5120 2 -1"),
    ("r06_pmc_gemm_ffndown.txt", "gemm_bf16_384ILi2E", ["gemm_14400x5120x13824_epi2"], 2 * (14400 * 13824 + 5120 * 13824 + 2 * 14400 * 5120),
     "gemm_bf16_384<EPI_GATE_RES> 14400 x 5120 x 13824 (FFN-down: bf16 operands, bf16 residual in, bf16 out)", "tools/one_kernel.py gemm 14400 5120 13824 2 -1"),
    ("r06_pmc_gemm8_ffndown.txt", "gemm_fp8_w4ILi2E", ["gemm_mxfp8_14400x5120x13824_epi2"],
     14400 * 13824 + 5120 * 13824 + 2 * 2 * 14400 * 5120 + (14400 + 5120) * 13824 // 32,
     "gemm_fp8_w4<EPI_GATE_RES, MX> 14400 x 5120 x 13824 (FFN-down: e4m3 operands, bf16 residual in, bf16 out)", "tools/one_kernel.py gemm8 14400 5120 13824 2"),
    ("r06_pmc_gemm_ffnup.txt", "gemm_bf16_w4ILi1E", ["gemm_14400x13824x5120_epi1"], 687144960,
     "gemm_bf16_w4<EPI_BIAS_GELU> (256 x 256 tile, one wave per SIMD; its split-K reduce launch not included) 14400 x 13824 x 5120", "tools/one_kernel.py gemm 14400 13824 5120 1 -1"),
    ("r05_pmc_attn8_7200_b2.txt", "attn_fwd_mxfp8_sp_kernel", ["attention_mxfp8_7200x7200_h40_b2", "attention_mxfp8_7200x7200_h40_b2_mxq"],
     2 * 7200 * 5120 * (1 + 1 + 1 + 2) + 3 * 2 * 7200 * 5120 // 32,
     "attn_fwd_mxfp8_sp_kernel, 7200 keys x 40 heads x 2 samples (q8 + k8 + v8t e4m3, bf16 output, E8M0 scales)", "tools/one_kernel.py attn8 7200 40 2"),
    ("r06_pmc_gemm8_outproj.txt", "gemm_fp8_w4ILi2E", ["gemm_mxfp8_14400x5120x5120_epi2"],
     14400 * 5120 + 5120 * 5120 + 2 * 2 * 14400 * 5120 + (14400 + 5120) * 5120 // 32,
     "gemm_fp8_w4<EPI_GATE_RES, MX> 14400 x 5120 x 5120 (e4m3 operands, bf16 residual in, bf16 out)", "tools/one_kernel.py gemm8 14400 5120 5120 2"),
    ("r06_pmc_gemm8_ffnup.txt", "gemm_fp8_w4ILi7E", ["gemm_mxfp8_14400x13824x5120_gelu_quant"],
     14400 * 5120 + 13824 * 5120 + 14400 * 13824 + (14400 * 5120 + 13824 * 5120 + 14400 * 13824) // 32,
     "gemm_fp8_w4<EPI_BIAS_GELU_Q, MX> 14400 x 13824 x 5120 (e4m3 operands, e4m3 + E8M0 output)", "tools/one_kernel.py gemm8 14400 13824 5120 7"),
]


def parse(path, frag):
    vals, dur = {}, None
    cur = None
    for ln in open(path):
        ln = ln.rstrip()
        m = re.match(r"^(p\d+) (\S.*)$", ln)
        if m:
            cur = frag in m.group(2)
            continue
        m = re.match(r"^duration_ns (.*) mean ([0-9.e+]+) n", ln)
        if m:
            if frag in m.group(1):
                dur = float(m.group(2))
            continue
        m = re.match(r"^\s+(\w+): mean ([0-9.e+-]+) over", ln)
        if m and cur:
            vals[m.group(1)] = float(m.group(2))
    return vals, dur


def main():
    out = {"_comment": "HBM-side traffic per launch of the dominant bf16 and fp8 kernels from rocprofv3 --pmc passes (one counter set per pass, --kernel-trace only; GEMMs: round 6 - the register-direct "
                       "epilogues - tools/sessions_r06/gpu_r6_i.sh, r06_pmc_*.txt; attention kernels: unchanged since round 5, r05_pmc_*.txt).  fetch_bytes = FETCH_SIZE x 1024 x 2 (gfx950 tallies "
                       "128-byte requests at 64), write_bytes = WRITE_SIZE x 1024; Infinity-Cache hits are included: traffic past the L2, an upper bound on HBM bytes.  "
                       "Shapes not listed fall back to r05 / r04 / r02_pmc_traffic.json (bench.py::_pmc_traffic)."}
    for fn, frag, labels, alg, kernel, cmd in SPECS:
        path = os.path.join(P, fn)
        if not os.path.exists(path):
            continue
        v, dur = parse(path, frag)
        if "FETCH_SIZE" not in v:
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        rec = {"fetch_bytes": int(v["FETCH_SIZE"] * 1024 * 2), "write_bytes": int(v["WRITE_SIZE"] * 1024), "algorithmic_bytes": int(alg), "kernel": kernel,
               "l2_hit_rate": round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3),
               "mfma_busy_frac": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 3),
               "effective_clock_GHz": None if not dur else round(cyc / dur, 3),
               "wave_time_split": {k: round(v[k] / v["SQ_WAVE_CYCLES"], 3) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in v},
               "lds_issue_stall_frac_of_wave_time": round(v.get("SQ_WAIT_INST_LDS", 0.0) / v["SQ_WAVE_CYCLES"], 4),
               "note": f"{cmd}; raw: {fn}; traffic / algorithmic = {(v['FETCH_SIZE'] * 2048 + v['WRITE_SIZE'] * 1024) / alg:.2f} x"}
        for lb in labels:
            out[lb] = rec
    json.dump(out, open(os.path.join(P, "r06_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
