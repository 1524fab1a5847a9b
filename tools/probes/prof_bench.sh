#!/bin/bash
# prof_bench.sh <substr> [bench args]: rocprofv3 kernel statistics of a short bench.py run, filtered by kernel name
sub=$1; shift
export TMPDIR=/tmp
d=$(mktemp -d /tmp/pb.XXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o s -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof-gate "$@" > $d/log.txt 2>&1
python3 tools/probes/kstats.py $d "$sub"
