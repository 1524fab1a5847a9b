cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
tools/ab_env.sh "--steps 60 --warmup 5" DF_STEM_NOZ=0 DF_STEM_NOZ=1 2>&1 | tee $O/ab.log
python -m pytest tests/test_hip_whole_step.py tests/test_hip_parity.py tests/test_error_budget.py tests/test_hip_fullsize.py tests/test_data_loader.py -m gpu -x -q 2>&1 | tail -5 | tee $O/tests.log
