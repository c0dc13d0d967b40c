#!/bin/bash
# SQ LDS / VALU counters per kernel of the default bench (one rocprofv3 --pmc pass, eager launches)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/raw -o lds -- python3 $ROOT/bench.py --no-cpu-baseline --no-kernel-classes --no-line-grid --no-other-configs --no-cond-table --steps 2 --warmup 1 --no-graph > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
python3 - "$f" > $OUT/summary.txt <<PY
import csv,sys,collections
agg=collections.defaultdict(lambda:collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0][:50]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0))[:8]:
    print(k, {c:round(x/1e6,1) for c,x in v.items()})
PY
rm -rf $OUT/raw
cat $OUT/summary.txt
