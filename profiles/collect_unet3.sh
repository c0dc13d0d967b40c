#!/bin/bash
# rocprofv3 evidence for unet3 (train_ultra_res.py:51-60: 96 % of a patch's work, the whole of configs[3] / [4]'s stage 3)
# at batch 1 (one patch of the ultra-res grid) and batch 8 (configs[3]), run from the repo root on the GPU box:
#   profiles/collect_unet3.sh <tag>   ->  gpurun_out/prof_<tag>/unet3_b{1,8}_{kernel_stats.csv,sq_summary.json,fwd.log}
# Two passes per batch: --kernel-trace --stats, and the SQ counters (--kernel-trace + --pmc only, as MI355X_MICROARCH.md
# prescribes).  The traced program is scratch/fwd_configs.py: plan build + 4 forwards + 2 per-op profile runs.
set -o pipefail
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$B -o stats -- python3 $ROOT/scratch/fwd_configs.py unet3 $B > $OUT/unet3_b${B}_fwd.log 2> $OUT/unet3_b${B}.err || exit 1
  for f in $(find $OUT/st$B -name "*kernel_stats.csv"); do cp $f $OUT/unet3_b${B}_kernel_stats.csv; done
  rm -rf $OUT/st$B
  echo "unet3 b$B stats pass done"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq$B -o sq -- python3 $ROOT/scratch/fwd_configs.py unet3 $B > /dev/null 2> $OUT/unet3_b${B}_sq.err || exit 1
  cc=$(find $OUT/sq$B -name "*counter_collection.csv" | head -1)
  kt=$(find $OUT/sq$B -name "*kernel_trace.csv" | head -1)
  python3 $ROOT/profiles/reduce_sq.py $cc $kt > $OUT/unet3_b${B}_sq_summary.json
  rm -rf $OUT/sq$B
  echo "unet3 b$B sq pass done"
done
ls -la $OUT
