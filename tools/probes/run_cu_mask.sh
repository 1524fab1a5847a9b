mkdir -p gpurun_out/r05b
./tools/probes/build_cu_mask_map > gpurun_out/r05b/cu_mask_map.txt 2>&1
for m in lo128 hi128 lo192 lo96 lo64; do
  DFHIP_CU_MASK=$m python bench.py --no-cpu-baseline --no-other-model > gpurun_out/r05b/bench_$m.json 2>gpurun_out/r05b/bench_$m.err
done
DFHIP_CU_MASK=lo128 python bench.py --no-cpu-baseline --no-other-model --no-graph > gpurun_out/r05b/bench_lo128_nograph.json 2>gpurun_out/r05b/bench_lo128_nograph.err
python bench.py --no-cpu-baseline --no-other-model > gpurun_out/r05b/bench_full.json 2>gpurun_out/r05b/bench_full.err
python -m pytest tests/test_hip_lazy_records.py -x -q > gpurun_out/r05b/lazy_tests.txt 2>&1
tail -30 gpurun_out/r05b/lazy_tests.txt
cat gpurun_out/r05b/cu_mask_map.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05b/bench_*.json')):
    try: d=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e: print(f,'failed',e); continue
    print(f.split('/')[-1], d['ms_per_step'])
PY
