cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "stem" 2>&1 | tail -12
python -m pytest tests/test_hip_whole_step.py tests/test_hip_parity.py tests/test_error_budget.py -m gpu -x -q 2>&1 | tail -4
tools/ab_env.sh "--workload cnn --steps 40 --warmup 5" DF_STEM_NOZ=0 DF_STEM_NOZ=1
tools/ab_env.sh "--steps 60 --warmup 5" DF_STEM_NOZ=0 DF_STEM_NOZ=1
