#!/bin/bash
# round 6, session v: evidence on the tree with the new VAE kernels - whole GPU suite (timed), smoke, the VAE account at 720p (fused / unfused
# norm) and at 1584x1056, rocprofv3 kernel stats of the VAE, the driver's bench command
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; O=gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu --durations=25 -x ) > $O/r6v_pytest.log 2>&1; tail -32 $O/r6v_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for f in 0 1 0 1; do
  echo "== CE_VAE_FUSE_NORM=$f" | tee -a $O/r6v_vae_fuse_ab.txt
  CE_VAE_FUSE_NORM=$f CE_VAE_BENCH_OUT=$O/r6v_vae_bench_720p_fuse$f.json timeout 600 python tools/vae_bench.py 2>&1 | grep "hipGraph replay" | tee -a $O/r6v_vae_fuse_ab.txt
done
timeout 900 python tools/vae_bench.py 1056 1584 5 2>&1 | grep -v amdgpu.ids | grep "encode\|decode\|max mem" | head -8 | tee $O/r6v_vae_bench_1584x1056.txt
(cd /tmp && CE_VAE_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r6v_prof_vae -o vae -- python $R/tools/vae_bench.py > $R/$O/r6v_rocprof_vae.log 2>&1)
f=$(find $O/r6v_prof_vae -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r6v_vae_kernel_stats.csv; rm -rf $O/r6v_prof_vae
head -12 $O/r6v_vae_kernel_stats.csv | cut -c1-200
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6v_bench.json 2> $O/r6v_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6v_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['mfma_roofline_frac_whole_step'], d['steps_per_sec_fp8_mode'], d['fp8_mode_frac_of_fp8_peak'], d['steps_per_sec_fp8_config4']['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_family']['frac'])
print({k: v['seconds'] for k, v in d['sec_per_edit'].items() if isinstance(v, dict)}, d['vae'], d['power'])
PY
