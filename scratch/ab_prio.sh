#!/bin/bash
# A/B of KD_FWINO_PRIO variants on ONE box with the experiment library (lib_x): per-op profiles of the headline UNet
out=gpurun_out/$1; shift
mkdir -p $out
export KD_ENGINE_LIB=$PWD/kidney-diffusion_amd/lib_x/libkd_engine.so
for v in "$@"; do
  KD_FWINO_PRIO=$v python scratch/dump_ops.py 16 > $out/ops_prio$v.csv 2> $out/ops_prio$v.err || { tail -5 $out/ops_prio$v.err; exit 1; }
done
python scratch/ops_summary.py $(for v in "$@"; do echo $out/ops_prio$v.csv; done) | grep -E "==|fused Winograd|wino fused"
