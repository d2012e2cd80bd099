#!/bin/bash
# round 5, session k: the profile refresh - the driver's bench command, rocprofv3 kernel stats of the bf16 and the fp8 step, the other BASELINE
# shapes, PMC passes of the two fp8 GEMM forms on the new LDS-DMA schedule
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5k_bench.json 2> $O/r5k_bench.err; echo "bench rc $?"; tail -c 600 $O/r5k_bench.json
F="--no-cpu-baseline --no-profile --no-vae --no-encoders --no-fp8-leg --no-edit"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r5k_prof -o bench -- python $R/bench.py --steps 5 --warmup 1 $F > $R/$O/r5k_rocprof.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r5k_prof_fp8 -o bench -- python $R/bench.py --steps 5 --warmup 1 --fp8 $F > $R/$O/r5k_rocprof_fp8.log 2>&1)
find $O/r5k_prof $O/r5k_prof_fp8 -name "*kernel_stats.csv" | head
for d in r5k_prof r5k_prof_fp8; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
bash tools/gpu_configs.sh > $O/r5k_configs.txt 2>&1; cp $O/configs.log $O/r5k_configs.log; tail -14 $O/r5k_configs.txt
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
pmc r5k_gemm8_outproj gemm8 14400 5120 5120 2 6
pmc r5k_gemm8_ffnup gemm8 14400 13824 5120 7 6
grep -A3 "SQ_VALU_MFMA_BUSY\|GRBM_GUI" $O/pmc_r5k_gemm8_outproj.txt | head -12
