#!/bin/bash
# same-box A/B of the 128-row tile of gemm_bf16x3_kernel (experiment build in lib_x/)
export KD_ENGINE_LIB=$PWD/kidney-diffusion_amd/lib_x/libkd_engine.so
run() {
  env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --no-line-grid --no-other-configs --no-kernel-classes --steps 20 --warmup 5 > gpurun_out/b_ab.json 2> gpurun_out/b_ab.err
  python -c "import json,sys;d=json.load(open('gpurun_out/b_ab.json'));print(' '.join(sys.argv[1:]), round(d['ms_per_step'],3))" "$@"
}
for rep in 1 2; do
  run KD_X3_BM128=0
  run KD_X3_BM128=1
  run KD_X3_BM128=1 KD_X3_BM128_MAXK=256
  run KD_X3_BM128=1 KD_X3_BM128_MAXK=128
done
