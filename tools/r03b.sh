cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests/test_data_loader.py tests/test_hip_fullsize.py tests/test_hip_parity.py tests/test_hip_edge_cases.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.log
for v in 0 1 0 1 8 2; do echo "== DFHIP_XCD_ORDER=$v"; DFHIP_XCD_ORDER=$v python tools/conv_bench.py --only l3,l4,l4a,l3a --iters 200 2>&1 | grep -E "fwd_stats|dgrad_ow|dgrad_pair|fwd " ; done > $O/xcd_convbench.log 2>&1
tools/ab_env.sh "--steps 60 --warmup 5" DFHIP_XCD_ORDER=0 DFHIP_XCD_ORDER=1 > $O/xcd_step.log 2>&1
for v in 0 1; do
  DFHIP_XCD_ORDER=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$v -o p -- python3 tools/conv_bench.py --only l3,l4,l4a,l3a --iters 20 > $O/pmc_$v.log 2>&1
  python3 tools/pmc_summary.py $(find $O/pmc_fetch_$v -name "*counter_collection.csv") > $O/pmc_per_kernel_$v.txt 2>&1
  rm -rf $O/pmc_fetch_$v
done
tail -5 $O/tests.log; cat $O/xcd_step.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k wide 2>&1 | tail -15 > $O/tests_wide.log
timeout 900 python bench.py --workload resnet18_wide --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_wide.json 2> $O/bench_wide.err
tail -3 $O/tests_wide.log; tail -c 400 $O/bench_wide.json; tail -3 $O/bench_wide.err
