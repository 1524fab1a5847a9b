mkdir -p gpurun_out/rider
DFHIP_LIBRARY=tools/probes/build/rider/libdfhip.so DFHIP_RIDER_WGS=256 DFHIP_RIDER_ITERS=2000 python bench.py --no-cpu-baseline --no-other-model --sustain-ms 0 --steps 5 > gpurun_out/rider/dbg.json 2> gpurun_out/rider/dbg.err
grep rider gpurun_out/rider/dbg.err | head; python -c "
import json; d=json.loads(open('gpurun_out/rider/dbg.json').read().strip().split('\n')[-1]); print(d['ms_per_step']); print([(r['kernel'][:50],r['avg_us']) for r in d['kernel_table'][:6]])"
