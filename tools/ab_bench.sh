#!/bin/bash
# A/B two builds of libdfhip.so on the SAME box (box-to-box variance is ~1 %): alternates them,
# prints ms per step of the captured ResNet-18 step.   usage: tools/ab_bench.sh <a.so> <b.so> [rounds]
set -e
A=$1; B=$2; R=${3:-3}
cp deepflows_amd/libdfhip.so /tmp/libdfhip_orig.so
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then cp $A deepflows_amd/libdfhip.so; else cp $B deepflows_amd/libdfhip.so; fi
    python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['value']))"
  done
done
cp /tmp/libdfhip_orig.so deepflows_amd/libdfhip.so
