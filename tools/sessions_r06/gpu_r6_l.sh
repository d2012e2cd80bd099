#!/bin/bash
# round 6, session l: PMC passes of the FFN-down launches (the largest label of the dominant symbol gemm_bf16_384<2>, and its fp8 twin)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
O=gpurun_out
pmc() {
  tag=$1; shift
  mkdir -p $O/pmc_$tag
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
  rm -rf $O/pmc_$tag/p*/
}
pmc r6l_gemm_ffndown gemm 14400 5120 13824 2 -1 6
pmc r6l_gemm8_ffndown gemm8 14400 5120 13824 2 6
grep -c "mean" $O/pmc_r6l_*.txt
