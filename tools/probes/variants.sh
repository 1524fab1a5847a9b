# tools/probes/variants.sh <tag> <kernel filter> <variant> [<variant> ...]: the headline bench with each diagnostic build
# (tools/build_variant.sh) and with the product library; prints the filtered kernels' durations (timing only)
TAG=$1; FILT=$2; shift 2
mkdir -p gpurun_out/$TAG
for v in "$@" product; do
  if [ $v = product ]; then python bench.py --no-cpu-baseline --no-other-model > gpurun_out/$TAG/$v.json 2>/dev/null
  else DFHIP_LIBRARY=tools/probes/build/$v/libdfhip.so python bench.py --no-cpu-baseline --no-other-model > gpurun_out/$TAG/$v.json 2>/dev/null; fi
done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/$TAG/*.json')):
    try: d=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e: print(f, 'failed', e); continue
    kt=d.get('kernel_table') or []
    rows=[(r.get('kernel','')[:44], r.get('avg_us'), r.get('launches_per_step')) for r in kt if any(s in r.get('kernel','') for s in '$FILT'.split(','))]
    print(f.split('/')[-1], round(d['ms_per_step'],4), rows)
PY
