#!/bin/bash
# One round's measurement set on the GPU box (run through gpurun): the bench lines of the three workloads and of
# the accumulate mode, the rocprofv3 kernel statistics of the default command, and the PMC passes (HBM traffic,
# matrix-pipe busy) summarised per kernel family.  Everything lands in gpurun_out/<tag>/; copy what is to be
# judged into profiles/.
#   usage: tools/profile_round.sh <tag> [steps]
set -u
TAG=${1:?tag}; STEPS=${2:-20}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --dgrad accumulate --no-cpu-baseline > $OUT/bench_accumulate.json 2> $OUT/bench_accumulate.err
python3 bench.py --workload cnn > $OUT/bench_cnn.json 2> $OUT/bench_cnn.err
python3 bench.py --workload mlp --steps 200 --warmup 20 > $OUT/bench_mlp.json 2> $OUT/bench_mlp.err
CMD="python3 bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-prof-gate"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT/stats.log 2>&1
for wl in cnn mlp; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -o s -- python3 bench.py --workload $wl --steps $STEPS --warmup 3 --no-cpu-baseline --no-prof-gate > $OUT/stats_$wl.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_acc -o s -- $CMD --dgrad accumulate > $OUT/stats_acc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
# steps the profiled command runs each family: warm-up 3 + 2 warm replays + STEPS replays + STEPS eager
N=$((3 + 2 + STEPS + STEPS))
python3 tools/pmc_summary.py --json $N $OUT/pmc_traffic.json $(find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq -name "*counter_collection.csv") > $OUT/pmc_summary.log 2>&1
python3 tools/pmc_summary.py $(find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection.csv") > $OUT/pmc_per_kernel.txt 2>&1
for d in stats stats_cnn stats_mlp stats_acc; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 tools/kernel_stats_txt.py $f > $OUT/${d}_kernel_stats.txt
done
python3 tools/roofline_table.py $OUT/bench.json > $OUT/roofline_table.md 2>/dev/null
ls -la $OUT | head -40
