#!/bin/bash
# per-op profiles of the headline UNet: baseline library (lib_x, built from the previous sources) vs the in-tree library
out=gpurun_out/$1; mkdir -p $out
for r in 1 2; do
  KD_ENGINE_LIB=$PWD/kidney-diffusion_amd/lib_x/libkd_engine.so python scratch/dump_ops.py 16 > $out/ops_base$r.csv 2>/dev/null
  python scratch/dump_ops.py 16 > $out/ops_new$r.csv 2>/dev/null
done
python scratch/ops_summary.py $out/ops_base1.csv $out/ops_new1.csv $out/ops_base2.csv $out/ops_new2.csv | grep -E "==|fused Winograd|wino fused"
