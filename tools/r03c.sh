cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
# what does a long matrix-bound launch on a side stream cost the backward chain inside the captured step?
# iters: 4 MFMAs x 32 cycles = 128 cycles per iteration; 2350 iterations ~ 0.15 ms at 2 GHz
tools/ab_env.sh "--steps 60 --warmup 5" DF_SIDE_PROBE= DF_SIDE_PROBE=64,2350,150000 DF_SIDE_PROBE=96,2350,150000 DF_SIDE_PROBE=128,2350,150000 DF_SIDE_PROBE=96,4700,150000 DF_SIDE_PROBE=128,4700,150000 DF_SIDE_PROBE=96,2350,0 DF_SIDE_PROBE=8,100,0 > $O/side_probe.log 2>&1
cat $O/side_probe.log
python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | tail -5 > $O/tests.log; cat $O/tests.log
