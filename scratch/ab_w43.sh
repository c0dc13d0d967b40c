#!/bin/bash
# per-op profiles of the headline UNet with and without Winograd F(4x4,3x3) (plan option wino43_min_cin)
out=gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do
  KD_W43=$v python scratch/dump_ops.py 16 > $out/ops_w43_$v.csv 2> $out/ops_w43_$v.err || { tail -5 $out/ops_w43_$v.err; exit 1; }
done
python scratch/ops_summary.py $(for v in "$@"; do echo $out/ops_w43_$v.csv; done) | grep -E "==|fused Winograd|wino|gn stats|gn fold"
