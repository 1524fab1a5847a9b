mkdir -p gpurun_out/r05j
python -m pytest tests -m gpu -q > gpurun_out/r05j/gpu_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r05j/gpu_tests.txt | tail -3
grep -n "^FAILED\|^ERROR" gpurun_out/r05j/gpu_tests.txt | cut -c1-200
export DFHIP_KERNEL_LOG=$PWD/gpurun_out/r05j/kernels.log
rm -f $DFHIP_KERNEL_LOG
python -m pytest tests -m gpu -q --deselect tests/test_hip_switches.py > gpurun_out/r05j/gpu_tests_klog.txt 2>&1
grep -n "passed\|failed" gpurun_out/r05j/gpu_tests_klog.txt | tail -3
B="python bench.py --no-cpu-baseline --sustain-ms 0 --steps 5"
$B --dgrad accumulate > gpurun_out/r05j/b2.json 2>gpurun_out/r05j/b2.err
$B --workload cnn > gpurun_out/r05j/b3.json 2>gpurun_out/r05j/b3.err
$B --workload mlp > gpurun_out/r05j/b4.json 2>gpurun_out/r05j/b4.err
$B --workload resnet18_wide > gpurun_out/r05j/b5.json 2>gpurun_out/r05j/b5.err
$B --epoch --epoch-images 5000 > gpurun_out/r05j/b7.json 2>gpurun_out/r05j/b7.err
DF_FORCE_COMM=1 DF_FORCE_SYNC=1 $B > gpurun_out/r05j/b8.json 2>gpurun_out/r05j/b8.err
$B --batch 128 > gpurun_out/r05j/b9.json 2>gpurun_out/r05j/b9.err
unset DFHIP_KERNEL_LOG
python tools/kernel_coverage.py gpurun_out/r05j/kernels.log > gpurun_out/r05j/coverage.txt 2>&1
head -3 gpurun_out/r05j/coverage.txt
python bench.py --no-cpu-baseline > gpurun_out/r05j/bench.json 2>gpurun_out/r05j/bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r05j/bench.json').read().strip().split('\n')[-1]); print({k:d[k] for k in ('value','ms_per_step','sustained_ms_per_step','median_ms_per_step','best_ms_per_step')})"
for f in gpurun_out/r05j/b*.err; do echo $f; grep -v "^$" $f | tail -2 | cut -c1-300; done
