mkdir -p gpurun_out/r05r
python -m pytest tests/test_hip_fullsize.py tests/test_hip_parity.py -q -x > gpurun_out/r05r/tests.txt 2>&1; tail -2 gpurun_out/r05r/tests.txt
B="python bench.py --no-cpu-baseline --no-other-model --sustain-ms 0"
for i in 1 2; do
DFHIP_LIBRARY=tools/probes/build/la1/libdfhip.so $B > gpurun_out/r05r/bench_old$i.json 2>gpurun_out/r05r/old.err
$B > gpurun_out/r05r/bench_new$i.json 2>gpurun_out/r05r/new.err
done
$B --dgrad accumulate > gpurun_out/r05r/bench_newacc.json 2>gpurun_out/r05r/newacc.err
DFHIP_LIBRARY=tools/probes/build/la1/libdfhip.so $B --dgrad accumulate > gpurun_out/r05r/bench_oldacc.json 2>gpurun_out/r05r/oldacc.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05r/bench_*.json')):
    try: d=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e: print(f,'failed',e); continue
    kt={r['kernel']:r['avg_us'] for r in d['kernel_table']}
    print(f.split('/')[-1], 'ms/step %.4f'%d['ms_per_step'], {k[:36]:v for k,v in kt.items() if 'img2_all' in k})
PY
