# Round 5: what a matrix-bound rider inside a chain kernel's launch costs and buys (diagnostic build -DDFHIP_RIDER_PROBE of
# dfhip_dgrad_ow.hip: DFHIP_RIDER_WGS extra workgroups of the 32-channel BatchNorm-form input gradients spin on
# DFHIP_RIDER_ITERS x 4 fp32 MFMAs per wave).  Prints the kernels' own durations from the bench line's kernel table.
#   ONLY=dfhip_dgrad_ow tools/build_variant.sh rider -DDFHIP_RIDER_PROBE;  gpurun -- bash tools/probes/rider_probe.sh
mkdir -p gpurun_out/rider
B="python bench.py --no-cpu-baseline --no-other-model --sustain-ms 0"
export DFHIP_LIBRARY=tools/probes/build/rider/libdfhip.so
for wgs in 0 256; do
  for it in 0 12 25 50 100 200 -60 -120 -240; do
    [ $wgs = 0 ] && [ $it != 0 ] && continue
    DFHIP_RIDER_WGS=$wgs DFHIP_RIDER_ITERS=$it $B > gpurun_out/rider/b_${wgs}_${it}.json 2> gpurun_out/rider/b_${wgs}_${it}.err
  done
done
python - <<'PY'
import json,glob,re
rows=[]
for f in glob.glob('gpurun_out/rider/b_*.json'):
    m=re.search(r'b_(\d+)_(-?\d+)\.json',f); wgs,it=int(m.group(1)),int(m.group(2))
    try: d=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e: print(f,'failed',e); continue
    kt={r['kernel']:r['avg_us'] for r in d['kernel_table']}
    rows.append((wgs,it,d['ms_per_step'],kt.get('k_dgrad_ow_s1<32, 0, false, true, 1, 8>'),kt.get('k_dgrad_ow_s1<64, 0, false, true, 0, 4>'),kt.get('k_dgrad_ow_s1<64, 0, false, false, 1, 4>')))
print('rider wgs  iters  rider us per SIMD (iters > 0: MFMA, 2 waves x iters x 4 x 32 cycles @2.2GHz; iters < 0: a sleeping rider of -iters x 0.1 us)  ms/step   us <32, 0, false, true, 1, 8> (160 VGPRs x 8 waves: one workgroup per CU)   us <64, 0, false, true, 0, 4>   us <64, 0, false, false, 1, 4> (4 waves)')
for wgs,it,ms,a,b,c in sorted(rows):
    print('%9d %6d %10.1f %50.4f %12s %16s %16s'%(wgs,it, (2*it*4*32/2200.0 if (wgs and it > 0) else (-it*0.1 if wgs else 0.0)), ms, a, b, c))
PY
