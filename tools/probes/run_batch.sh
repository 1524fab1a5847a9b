mkdir -p gpurun_out/r05k
B="python bench.py --no-cpu-baseline --no-other-model --sustain-ms 1000"
for v in default HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0; do
  if [ $v = default ]; then $B > gpurun_out/r05k/bench_$v.json 2>gpurun_out/r05k/$v.err; else env $v $B > gpurun_out/r05k/bench_$v.json 2>gpurun_out/r05k/$v.err; fi
done
python - <<'PY'
import json,glob
base=None
for f in sorted(glob.glob('gpurun_out/r05k/bench_*.json')):
    try: d=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e: print(f,'failed',e); continue
    kt={r['kernel']:r['avg_us'] for r in d['kernel_table']}
    print(f.split('/')[-1], 'ms/step %.4f sustained %.4f best %.4f'%(d['ms_per_step'], d['sustained_ms_per_step'] or 0, d['best_ms_per_step'] or 0), 'adam %.1f slabsum %.1f'%(kt.get('k_adam',0), kt.get('k_wgrad_group_sum',0)), 'sum of kernels %.1f us'%sum(r['avg_us']*r['launches_per_step'] for r in d['kernel_table']))
PY
