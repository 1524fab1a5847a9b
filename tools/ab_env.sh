#!/bin/bash
# A/B of environment switches on ONE box: tools/ab_env.sh "<bench args>" VAR=v1 VAR=v2 ...  (each setting run twice, interleaved)
ARGS=$1; shift
for round in 1 2; do
  for kv in "$@"; do
    env $kv python3 bench.py --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$kv', round(d['value']), round(d['ms_per_step'],4), d['config']['dispatch'], round(d['final_loss'],6))"
  done
done
