cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "stem" 2>&1 | tail -25 > $O/tests.log; cat $O/tests.log
python -m pytest tests/test_hip_whole_step.py tests/test_hip_parity.py tests/test_error_budget.py -m gpu -x -q 2>&1 | tail -8 > $O/tests2.log; cat $O/tests2.log
tools/ab_env.sh "--steps 60 --warmup 5" DF_STEM_NOZ=0 DF_STEM_NOZ=1 > $O/ab.log 2>&1; cat $O/ab.log
python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03e/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['mfma_clock_mhz_under_load'], d['roofline']['mfma_probe_tflops'])
for k,v in d['kernels'].items(): print(k, v['calls_per_step'], v['kernels_per_call'], round(v['avg_us'],1), v['bound'], round(v['frac'],3), round(v['ms_per_step'],4))
PY
