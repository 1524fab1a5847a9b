cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_hip_edge_cases.py -m gpu -x -q -k "linear or mlp or gemm or head or vgg or cnn" 2>&1 | tail -4
for v in 0 1 0 1; do DF_LINEAR_RELU=$v python bench.py --workload mlp --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('DF_LINEAR_RELU=$v', round(d['value']), round(d['ms_per_step'],4), [(r['kernel'][:28], r['launches_per_step'], r['avg_us']) for r in d['kernel_table']])"; done
python bench.py --workload cnn --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cnn', round(d['value']), round(d['ms_per_step'],4))"
