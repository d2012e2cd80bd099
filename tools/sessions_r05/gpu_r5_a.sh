#!/bin/bash
# round 5, session a: the new bench legs on hardware (bare --gpus 2 self-launch under the test backend, cpu_baseline kind reference, configs[4]
# leg), the fp8 PMC passes asked for since round 3 (MX GEMM out-proj / FFN-up / FFN-down forms, MXFP8 attention), the fp8 MFMA ceiling probe
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
( timeout 900 python -m pytest tests/test_bench_multirank_gpu.py tests/test_dit_forward_gpu.py -x -q --durations=15 > $O/r5a_pytest.log 2>&1; echo "pytest rc $?" >> $O/r5a_pytest.log ) 
tail -25 $O/r5a_pytest.log
timeout 120 tools/probes/mfma_rate_probe_fp8 > $O/r5a_mfma_rate_probe_fp8.txt 2>&1; cat $O/r5a_mfma_rate_probe_fp8.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-reasoning-edit --no-full-edit > $O/r5a_bench.json 2> $O/r5a_bench.err; echo "bench rc $?"; tail -c 3000 $O/r5a_bench.json
# PMC: MX fp8 GEMMs of the step (q|k: 14400x10240x5120 epi0; out-proj 14400x5120x5120 epi2; FFN-up 14400x13824x5120 epi7; FFN-down 14400x5120x13824 epi2)
pmc() {  # tag, one_kernel args...
  tag=$1; shift
  mkdir -p $O/pmc_$tag
  R=$GRAFT_REPO_ROOT
  ( cd /tmp
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$O/pmc_$tag/p$i -o p -- python $R/tools/one_kernel.py "$@" > $R/$O/pmc_$tag/p$i.log 2>&1
  done )
  python - > $O/pmc_$tag.txt <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/pmc_$tag/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "at::" in name or "rocclr" in name or "quant" in name or "transpose" in name or "rmsnorm" in name: continue
        agg[name[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(f.split("/")[2], k)
        for c, v in d.items():
            print(f"   {c}: mean {sum(v)/len(v):.6g} over {len(v)}")
for f in sorted(glob.glob("$O/pmc_$tag/p1/**/*kernel_trace.csv", recursive=True)):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in d.items():
        if "at::" in k: continue
        print("duration_ns", k, "mean", sum(v)/len(v), "n", len(v))
PY
  cat $O/pmc_$tag.txt
}
pmc r5_gemm8_outproj gemm8 14400 5120 5120 2 6
pmc r5_gemm8_ffnup gemm8 14400 13824 5120 7 6
pmc r5_attn8_7200_b2 attn8 7200 40 2 6
rm -rf $O/pmc_*/p*/  # the raw per-pass directories are large; the summaries stay
