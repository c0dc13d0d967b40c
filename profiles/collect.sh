#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's default workload on the GPU box (run from the repo root):
#   profiles/collect.sh <tag>        ->  gpurun_out/prof_<tag>/{kernel_stats.csv, pmc_fetch.csv, pmc_write.csv, sq.csv, sq_trace.csv}
# Four separate passes: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; SQ counters (each PMC pass with
# --kernel-trace only, as MI355X_MICROARCH.md prescribes).  The program after `--` is python3 itself.
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --no-cond-table: the table of the time conditioning is built once per schedule (250 x 21 small launches + 28 GB of weight
# reads); in a 4-7 step trace that one-off work would be booked as per-step time and traffic
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-kernel-classes --no-line-grid --no-other-configs --no-cond-table"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH --steps 5 --warmup 2 > $OUT/stats_bench.json 2> $OUT/stats.err || exit 1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $BENCH --steps 3 --warmup 1 --no-graph > $OUT/fetch_bench.json 2> $OUT/fetch.err || exit 1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $BENCH --steps 3 --warmup 1 --no-graph > $OUT/write_bench.json 2> $OUT/write.err || exit 1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o sq -- $BENCH --steps 3 --warmup 1 --no-graph > $OUT/sq_bench.json 2> $OUT/sq.err || exit 1
echo "sq pass done"
find $OUT -name "*.csv" | head -20
# keep only the per-kernel tables (the raw traces are large): gpurun merges <= 64 MiB back
for f in $(find $OUT -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
for f in $(find $OUT/fetch -name "*counter_collection.csv"); do cp $f $OUT/pmc_fetch.csv; done
for f in $(find $OUT/write -name "*counter_collection.csv"); do cp $f $OUT/pmc_write.csv; done
for f in $(find $OUT/sq -name "*counter_collection.csv"); do cp $f $OUT/sq.csv; done
for f in $(find $OUT/sq -name "*kernel_trace.csv"); do cp $f $OUT/sq_trace.csv; done
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq
python3 $ROOT/profiles/reduce_sq.py $OUT/sq.csv $OUT/sq_trace.csv > $OUT/sq_summary.json && rm -f $OUT/sq.csv $OUT/sq_trace.csv
ls -la $OUT
