mkdir -p gpurun_out/r04h
python -m pytest tests -m gpu -q -x > gpurun_out/r04h/tests.txt 2>&1
for i in 1 2 3; do
DFHIP_IMG4_HALVES=0 python bench.py --no-cpu-baseline --no-other-model > gpurun_out/r04h/off$i.json 2>/dev/null
python bench.py --no-cpu-baseline --no-other-model > gpurun_out/r04h/on$i.json 2>/dev/null
done
grep -E "passed|failed|error" gpurun_out/r04h/tests.txt | tail -3
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04h/*.json')):
    d=json.loads(open(f).read().strip().split('\n')[-1])
    kt=d.get('kernel_table') or d.get('config',{}).get('kernel_table') or []
    rows=[(r.get('kernel','')[:40], r.get('avg_us'), r.get('launches_per_step')) for r in kt if 'img4' in r.get('kernel','')] if isinstance(kt,list) else []
    print(f.split('/')[-1], round(d['ms_per_step'],4), rows)
PY
