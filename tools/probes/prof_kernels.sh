#!/bin/bash
# prof_kernels.sh <substr> <conv_bench args...>: rocprofv3 kernel statistics of a conv_bench run, filtered by name
sub=$1; shift
export TMPDIR=/tmp
d=$(mktemp -d /tmp/pk.XXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o s -- python3 tools/conv_bench.py "$@" > $d/log.txt 2>&1
python3 tools/probes/kstats.py $d "$sub"
