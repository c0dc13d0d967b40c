#!/bin/bash
# A/B/C of the bf16x3 position GEMMs on one box, interleaved twice: default plan, bf16x3 with the old Cin >= 512 rule, fp32 MFMA GEMMs
mkdir -p gpurun_out/abx3
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-line-grid --no-kernel-classes --steps 40 > gpurun_out/abx3/a_$rep.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-line-grid --no-kernel-classes --steps 40 --wino43-min-cin 512 > gpurun_out/abx3/b_$rep.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-line-grid --no-kernel-classes --steps 40 --fp32-mfma-gemms > gpurun_out/abx3/c_$rep.json 2>/dev/null
done
python - <<EOF2
import json,glob
for f in sorted(glob.glob("gpurun_out/abx3/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), round(d["value"],3))
EOF2
