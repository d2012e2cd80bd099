#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass, kernel-trace only) + L2 hit for ONE kernel shape.
# usage: gpu_pmc_traffic.sh <tag> <one_kernel.py args...>   -> gpurun_out/pmc_<tag>.txt
tag=$1; shift
mkdir -p gpurun_out/pmc_$tag
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -o p -- python $R/tools/one_kernel.py "$@" > $R/gpurun_out/pmc_$tag/p$i.log 2>&1
done
cd $R
python - > gpurun_out/pmc_$tag.txt <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_$tag/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "at::" in name or "rocclr" in name: continue
        agg[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(f.split("/")[2], k)
        for c, v in d.items():
            print(f"   {c}: mean {sum(v)/len(v):.6g} over {len(v)}")
PY
cat gpurun_out/pmc_$tag.txt
