#!/bin/bash
# round 6, session i: profile refresh on the register-direct epilogues - PMC passes of the four changed GEMM forms (bf16 out-projection on the 384-row
# tile, bf16 FFN-up, MX fp8 out-projection, MX fp8 FFN-up), energy per flop of the step's kernels, rocprofv3 kernel stats of configs[1] / fp8 / configs[2]
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
pmc r6i_gemm_outproj gemm 14400 5120 5120 2 -1 6
pmc r6i_gemm_ffnup gemm 14400 13824 5120 1 -1 6
pmc r6i_gemm8_outproj gemm8 14400 5120 5120 2 6
pmc r6i_gemm8_ffnup gemm8 14400 13824 5120 7 6
grep -c "mean" $O/pmc_r6i_*.txt
timeout 600 python tools/kernel_power.py 2>&1 | grep -v amdgpu.ids | tee $O/r6i_kernel_energy.txt
F="--no-cpu-baseline --no-profile --no-vae --no-encoders --no-fp8-leg --no-edit --no-full-edit --no-reasoning-edit"
for c in "c1:" "fp8:--fp8" "c2:--guidance 1.0 --steps 8"; do
  tag=${c%%:*}; args=${c#*:}
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r6i_prof_$tag -o bench -- python $R/bench.py --steps 5 --warmup 1 $F $args > $R/$O/r6i_rocprof_$tag.log 2>&1)
  f=$(find $O/r6i_prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r6i_prof_${tag}_kernel_stats.csv; rm -rf $O/r6i_prof_$tag
done
ls -la $O/r6i_*
