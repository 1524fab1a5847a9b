cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_whole_step.py -m gpu -x -q -k "optim or adam or whole or captured or resnet18_one" 2>&1 | tail -5 > $O/tests.log; cat $O/tests.log
python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03d/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['mfma_clock_mhz_under_load'], d['roofline']['mfma_probe_tflops'])
for k,v in d['kernels'].items(): print(k, v['calls_per_step'], round(v['avg_us'],1), v['bound'], round(v['frac'],3), round(v['ms_per_step'],4))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wide -o s -- python3 bench.py --workload resnet18_wide --steps 5 --warmup 2 --no-cpu-baseline --no-prof-gate > $O/stats_wide.log 2>&1
f=$(find $O/stats_wide -name "*kernel_stats.csv" | head -1); python3 tools/kernel_stats_txt.py $f > $O/wide_kernel_stats.txt; rm -rf $O/stats_wide; head -40 $O/wide_kernel_stats.txt
