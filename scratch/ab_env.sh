#!/bin/bash
# A/B of one experiment switch on ONE box with the experiment library (lib_x): per-op profiles of the headline UNet
#   scratch/ab_env.sh <outdir> <ENVVAR> <value> [<value> ...]
out=gpurun_out/$1; var=$2; shift 2
mkdir -p $out
export KD_ENGINE_LIB=$PWD/kidney-diffusion_amd/lib_x/libkd_engine.so
for v in "$@"; do
  env $var=$v python scratch/dump_ops.py 16 > $out/ops_${var}_$v.csv 2> $out/ops_${var}_$v.err || { tail -5 $out/ops_${var}_$v.err; exit 1; }
done
python scratch/ops_summary.py $(for v in "$@"; do echo $out/ops_${var}_$v.csv; done) | grep -E "==|fused Winograd|wino fused|wino#_in|wino#_out|wino# gemm"
