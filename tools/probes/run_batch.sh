mkdir -p gpurun_out/r05p
DFHIP_LIBRARY=tools/probes/build/s2c64/libdfhip.so python -m pytest tests/test_hip_parity.py -q -k "stride2" > gpurun_out/r05p/s2_tests.txt 2>&1; tail -3 gpurun_out/r05p/s2_tests.txt
B="python bench.py --no-cpu-baseline --no-other-model --sustain-ms 0"
for i in 1 2; do
$B > gpurun_out/r05p/bench_default$i.json 2>gpurun_out/r05p/default.err
DFHIP_LIBRARY=tools/probes/build/s2c64/libdfhip.so $B > gpurun_out/r05p/bench_v_$i.json 2>gpurun_out/r05p/v.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05p/bench_*.json')):
    try: d=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e: print(f,'failed',e); continue
    kt={r['kernel']:r['avg_us'] for r in d['kernel_table']}
    print(f.split('/')[-1], 'ms/step %.4f'%d['ms_per_step'], {k[:40]:v for k,v in kt.items() if 's2_pair<64' in k})
PY
