cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "stem" 2>&1 | tail -3
python tools/stem_bench.py 2>&1 | tee $O/stem_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python3 tools/stem_bench.py --iters 50 > $O/st.log 2>&1
f=$(find $O/st -name "*kernel_stats.csv" | head -1); python3 tools/kernel_stats_txt.py $f > $O/stem_kernel_stats.txt; rm -rf $O/st; cat $O/stem_kernel_stats.txt | head -14
