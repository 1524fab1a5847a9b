mkdir -p gpurun_out/r05a
for q in "" 8 16 2 1; do
  if [ -z "$q" ]; then ./tools/probes/build_two_streams2; else GPU_MAX_HW_QUEUES=$q ./tools/probes/build_two_streams2; fi
done > gpurun_out/r05a/two_streams2.txt 2>&1
HSA_ENABLE_INTERRUPT=0 ./tools/probes/build_two_streams2 > gpurun_out/r05a/two_streams2_noint.txt 2>&1
python bench.py --no-cpu-baseline --no-other-model > gpurun_out/r05a/bench.json 2>gpurun_out/r05a/bench.err
tail -c 600 gpurun_out/r05a/bench.json
cat gpurun_out/r05a/two_streams2.txt
