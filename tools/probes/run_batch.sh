mkdir -p gpurun_out/r05l
B="python bench.py --no-cpu-baseline --no-other-model --sustain-ms 0"
for i in 1 2; do
$B > gpurun_out/r05l/bench_default$i.json 2>gpurun_out/r05l/default.err
DFHIP_LIBRARY=tools/probes/build/rr9/libdfhip.so $B > gpurun_out/r05l/bench_rr9_$i.json 2>gpurun_out/r05l/rr9.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05l/bench_*.json')):
    try: d=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e: print(f,'failed',e); continue
    kt={r['kernel']:r['avg_us'] for r in d['kernel_table']}
    print(f.split('/')[-1], 'ms/step %.4f'%d['ms_per_step'], 'wgrad_group %.1f'%kt.get('k_wgrad_group',0), 'loss', d['final_loss'])
PY
