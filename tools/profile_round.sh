#!/bin/bash
# One round's measurement set on the GPU box (run through gpurun).  Per configuration (ResNet-18 overwrite = the
# headline, ResNet-18 accumulate, CNN; MLP and the wide ResNet without counters): the PMC passes FIRST (FETCH_SIZE,
# WRITE_SIZE, SQ counters: separate runs, --kernel-trace only) and their summary, then the rocprofv3 kernel statistics,
# then the bench line that quotes that summary, then the per-kernel table.  Everything lands in gpurun_out/<tag>/; copy
# what is to be judged into profiles/.
#   usage: tools/profile_round.sh <tag> [steps]
set -u
TAG=${1:?tag}; STEPS=${2:-20}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
# steps the profiled command runs each family: warm-up 3 + 2 warm replays + STEPS replays + STEPS eager
N=$((3 + 2 + STEPS + STEPS))
one() {   # one <suffix> <workload> <dgrad> [extra bench args]
  local sfx=$1 wl=$2 dg=$3; shift 3
  local extra="$*"
  local CMD="python3 bench.py --workload $wl --dgrad $dg --steps $STEPS --warmup 3 --no-cpu-baseline --no-prof-gate --no-other-model"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch$sfx -o p -- $CMD > $OUT/pmc_fetch$sfx.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write$sfx -o p -- $CMD > $OUT/pmc_write$sfx.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $OUT/pmc_sq$sfx -o p -- $CMD > $OUT/pmc_sq$sfx.log 2>&1
  python3 tools/pmc_summary.py --json $N $OUT/pmc_traffic$sfx.json --workload $wl --dgrad $dg \
      $(find $OUT/pmc_fetch$sfx $OUT/pmc_write$sfx $OUT/pmc_sq$sfx -name "*counter_collection.csv") > $OUT/pmc_summary$sfx.log 2>&1
  python3 tools/pmc_summary.py $(find $OUT/pmc_fetch$sfx $OUT/pmc_write$sfx $OUT/pmc_sq$sfx -name "*counter_collection.csv") > $OUT/pmc_per_kernel$sfx.txt 2>&1
  rm -rf $OUT/pmc_fetch$sfx $OUT/pmc_write$sfx $OUT/pmc_sq$sfx
  # where the waves' cycles go (quad-cycles): parked on s_waitcnt / barriers, stalled at issue, issuing; and which pipe issues
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES \
      --kernel-trace --output-format csv -d $OUT/pmc_wave$sfx -o p -- $CMD > $OUT/pmc_wave$sfx.log 2>&1
  python3 tools/pmc_any.py $(find $OUT/pmc_wave$sfx -name "*counter_collection.csv") > $OUT/pmc_wave_cycles$sfx.txt 2>&1
  rm -rf $OUT/pmc_wave$sfx
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$sfx -o s -- $CMD > $OUT/stats$sfx.log 2>&1
  f=$(find $OUT/stats$sfx -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 tools/kernel_stats_txt.py $f > $OUT/rocprof_kernel_stats$sfx.txt
  rm -rf $OUT/stats$sfx
  python3 bench.py --workload $wl --dgrad $dg --kernels-in-line --pmc-traffic $OUT/pmc_traffic$sfx.json $extra > $OUT/bench$sfx.json 2> $OUT/bench$sfx.err
  python3 tools/kernel_table.py $OUT/bench$sfx.json $OUT/pmc_traffic$sfx.json $OUT/rocprof_kernel_stats$sfx.txt > $OUT/kernel_table$sfx.md 2>> $OUT/bench$sfx.err
}
# CONFIGS="headline accumulate cnn rest" (default: all) picks the configurations; ONLY_HEADLINE=1 is CONFIGS=headline
CFG=${CONFIGS:-"headline accumulate cnn rest"}
[ -n "${ONLY_HEADLINE:-}" ] && CFG=headline
has() { case " $CFG " in *" $1 "*) return 0;; esac; return 1; }
has headline && one "" resnet18 overwrite
has accumulate && one _accumulate resnet18 accumulate --no-cpu-baseline
has cnn && one _cnn cnn overwrite
if has rest; then
python3 bench.py --workload mlp --steps 200 --warmup 20 --kernels-in-line > $OUT/bench_mlp.json 2> $OUT/bench_mlp.err
python3 bench.py --workload mlp --steps 200 --warmup 20 --steps-per-graph 4 --no-other-configs --no-cpu-baseline > $OUT/bench_mlp_spg4.json 2> $OUT/bench_mlp_spg4.err
python3 bench.py --matmul-dtype bf16 --no-cpu-baseline --no-other-configs --kernels-in-line > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
python3 bench.py --workload resnet18_wide --steps 10 --warmup 2 --no-cpu-baseline --kernels-in-line > $OUT/bench_wide.json 2> $OUT/bench_wide.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_wide -o s -- python3 bench.py --workload resnet18_wide --steps 5 --warmup 2 --no-cpu-baseline --no-prof-gate --no-other-model > $OUT/stats_wide.log 2>&1
f=$(find $OUT/stats_wide -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 tools/kernel_stats_txt.py $f > $OUT/rocprof_kernel_stats_wide.txt; rm -rf $OUT/stats_wide
fi
has headline && python3 tools/roofline_table.py $OUT/bench.json > $OUT/roofline_table.md 2>/dev/null
ls -la $OUT | head -60
