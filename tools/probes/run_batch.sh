mkdir -p gpurun_out/r05e
python -m pytest tests/test_hip_lazy_records.py -q > gpurun_out/r05e/lazy_tests.txt 2>&1
tail -15 gpurun_out/r05e/lazy_tests.txt
python bench.py --no-cpu-baseline --no-other-model --sustain-ms 0 > gpurun_out/r05e/bench_default.json 2>gpurun_out/r05e/bench_default.err
DFHIP_XCD_ORDER=1 python bench.py --no-cpu-baseline --no-other-model --sustain-ms 0 > gpurun_out/r05e/bench_xcd1.json 2>gpurun_out/r05e/bench_xcd1.err
python - <<'PY'
import json
def load(f): return json.loads(open(f).read().strip().split('\n')[-1])
a=load('gpurun_out/r05e/bench_default.json'); b=load('gpurun_out/r05e/bench_xcd1.json')
print('ms/step default %.4f xcd1 %.4f'%(a['ms_per_step'],b['ms_per_step']))
kb={r['kernel']:r for r in b['kernel_table']}
for r in a['kernel_table']:
    o=kb.get(r['kernel'])
    print('%-70s x%-3g %6.1f %6.1f'%(r['kernel'][:70], r['launches_per_step'], r['avg_us'], o['avg_us'] if o else -1))
PY
python -m pytest tests -m gpu -x -q > gpurun_out/r05e/gpu_tests.txt 2>&1
tail -8 gpurun_out/r05e/gpu_tests.txt
