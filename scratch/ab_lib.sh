#!/bin/bash
# same-box A/B of two builds of the engine: kidney-diffusion_amd/lib_old/libkd_engine.so (an earlier commit, see
# profiles/README.md) against the product library
run() {
  env "$@" python bench.py --no-cpu-baseline --no-line-grid --no-other-configs --no-kernel-classes --steps 20 --warmup 5 > gpurun_out/b_ab.json 2> gpurun_out/b_ab.err
  python -c "import json,sys;d=json.load(open('gpurun_out/b_ab.json'));print(' '.join(sys.argv[1:]), round(d['ms_per_step'],3))" "$@"
}
for rep in 1 2 3; do
  run KD_ENGINE_LIB=$PWD/kidney-diffusion_amd/lib_old/libkd_engine.so
  run KD_NEW=1
done
