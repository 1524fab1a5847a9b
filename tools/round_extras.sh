#!/bin/bash
# The round's measurements that are not part of tools/profile_round.sh (run on the GPU box through gpurun, after it):
#   soak.json            5 000 replays of the captured headline step, clock probe before / after (tools/soak.py)
#   exchange_world1.json bench.py with the exchange path forced on one GPU (RCCL communicator of one rank): exchange_captured,
#                        grads_copied_into_bucket, exchange_us_in_graph
#   epoch.json           bench.py --epoch: 50 000 images through DataLoader + device input pipeline + captured step
#   bn_bwd_bench.txt     the input-gradient kernels with / without the BatchNorm-backward forms (tools/bn_bwd_bench.py)
#   usage: tools/round_extras.sh <tag>
set -u
TAG=${1:?tag}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python3 tools/soak.py > $OUT/soak.json 2> $OUT/soak.err
DF_FORCE_COMM=1 DF_FORCE_SYNC=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29811 \
  python3 bench.py --no-cpu-baseline > $OUT/exchange_world1.json 2> $OUT/exchange_world1.err
python3 bench.py --epoch > $OUT/epoch.json 2> $OUT/epoch.err
python3 tools/bn_bwd_bench.py > $OUT/bn_bwd_bench.txt 2>&1
cut -c1-250 $OUT/bench.json 2>/dev/null
cut -c1-700 $OUT/soak.json
python3 -c "
import json,sys
d=json.loads(open('$OUT/exchange_world1.json').read().strip().split(chr(10))[-1]); print('exchange world 1:', d['ms_per_step'], {k: d['config'][k] for k in ('exchange_captured','grads_copied_into_bucket','exchange_us_in_graph','rccl_ranks')})"
cut -c1-420 $OUT/epoch.json
