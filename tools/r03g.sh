cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "stem" 2>&1 | tail -3
DFHIP_LIBRARY=tools/probes/build/stem2stamps/libdfhip.so python tools/stem_bench.py --only new_fwd --iters 40 2>&1 | tail -2
DFHIP_LIBRARY=tools/probes/build/stem2stamps/libdfhip.so python tools/stem_bench.py --only new_bwd --iters 40 2>&1 | tail -3
python tools/stem_bench.py
