# usage: tools/kstats.sh <out-prefix> [bench args...]  -- rocprofv3 kernel stats of a short bench run, top kernels as text
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=$1; shift
mkdir -p $(dirname $P)
rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_dir -o s -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-prof-gate "$@" > ${P}.log 2>&1
f=$(find ${P}_dir -name "*kernel_stats.csv" | head -1); python3 tools/kernel_stats_txt.py $f > ${P}_kernel_stats.txt; rm -rf ${P}_dir
