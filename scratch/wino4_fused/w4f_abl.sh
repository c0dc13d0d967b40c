#!/bin/bash
# ablations of the fused F(4x4,3x3) kernel with the experiment library (timing only): scratch/w4f_abl.sh <shape idx> <flags...>
export KD_ENGINE_LIB=$GRAFT_REPO_ROOT/kidney-diffusion_amd/lib_x/libkd_engine.so
shape=$1; shift
cd /tmp && export TMPDIR=/tmp
for f in "$@"; do
  export KD_W4F_FLAGS=$f
  rm -rf /tmp/w4f_abl
  rocprofv3 --kernel-trace --output-format csv -d /tmp/w4f_abl -o t -- python3 $GRAFT_REPO_ROOT/scratch/w4f_ab.py $shape > /tmp/w4f_abl.log 2>&1
  t=$(find /tmp/w4f_abl -name "*kernel_trace.csv" | head -1)
  echo "flags $f: $(python3 $GRAFT_REPO_ROOT/scratch/w4f_ab_parse.py $t)"
done
