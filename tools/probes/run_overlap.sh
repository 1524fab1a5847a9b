mkdir -p gpurun_out/r05c
O=gpurun_out/r05c/overlap_eager.txt
python tools/probes/overlap_eager.py > $O 2>gpurun_out/r05c/err0.txt
for side in 4 6 10; do
  for n in 96 128 160 192; do
    DFHIP_OVERLAP_CUS=$n DFHIP_WGRAD_SIDE=$side python tools/probes/overlap_eager.py >> $O 2>>gpurun_out/r05c/err1.txt
  done
done
DFHIP_WGRAD_SIDE=6 python tools/probes/overlap_eager.py >> $O 2>>gpurun_out/r05c/err1.txt
cat $O
python -m pytest tests/test_hip_lazy_records.py -q > gpurun_out/r05c/lazy_tests.txt 2>&1
tail -40 gpurun_out/r05c/lazy_tests.txt
python bench.py --no-other-model > gpurun_out/r05c/bench.json 2>gpurun_out/r05c/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r05c/bench.json').read().strip().split('\n')[-1]); print({k:d[k] for k in ('value','ms_per_step','sustained_ms_per_step','median_ms_per_step','best_ms_per_step','sustained')}); print(d['cpu_baseline'])"
