# same-box A/B of one environment switch on the headline bench: tools/probes/ab_env.sh <tag> <VAR=value> [kernel-name filter]
TAG=$1; SW=$2; FILT=${3:-rowmma}
mkdir -p gpurun_out/$TAG
for i in 1 2 3; do
env $SW python bench.py --no-cpu-baseline --no-other-model > gpurun_out/$TAG/off$i.json 2>/dev/null
python bench.py --no-cpu-baseline --no-other-model > gpurun_out/$TAG/on$i.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/$TAG/*.json')):
    d=json.loads(open(f).read().strip().split('\n')[-1])
    kt=d.get('kernel_table') or []
    rows=[(r.get('kernel','')[:44], r.get('avg_us'), r.get('launches_per_step')) for r in kt if any(s in r.get('kernel','') for s in '$FILT'.split(','))]
    print(f.split('/')[-1], round(d['ms_per_step'],4), len(kt), sum(r['launches_per_step'] for r in kt), rows)
PY
