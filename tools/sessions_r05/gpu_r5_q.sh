#!/bin/bash
# round 5, session q (final): the whole GPU suite + smoke on the final tree; fresh PMC traffic passes of the three dominant bf16 kernels
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
O=gpurun_out
( timeout 1500 python -m pytest tests -q -m gpu --durations=12 -x > $O/r5q_pytest.log 2>&1; echo "pytest rc $?" >> $O/r5q_pytest.log )
tail -22 $O/r5q_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
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
pmc r5q_attnvt_7200_b2 attnvt 7200 40 2 6
pmc r5q_gemm_outproj gemm 14400 5120 5120 2 -1 6
pmc r5q_gemm_ffnup gemm 14400 13824 5120 1 -1 6
grep -c "mean" $O/pmc_r5q_*.txt
