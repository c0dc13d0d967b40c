#!/usr/bin/env python3
"""Reduces two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE; each collected on its own with
--kernel-trace only, as MI355X_MICROARCH.md prescribes) of

    python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline

to HBM-side bytes per denoising step and writes profiles/hbm_traffic.json (read by bench.py for
`roofline.traffic`).

    python3 profiles/reduce_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> [steps=4]

gfx950 corrections: the counters are in KiB; FETCH_SIZE reports half of the bytes of wide coalesced
streaming reads, so it is doubled (cross-check printed below: gn_apply_silu reads and writes the same
tensor, its raw FETCH/WRITE ratio is 0.5).  Plan-build kernels (weight packing, copies) are excluded.
"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

# (split3_kernel: the once-per-plan split of the F(4x4,3x3) weights into bf16x3 planes - plan build, not a step)
EXCLUDE = ("pack_", "copyBuffer", "fillBuffer", "wino_pack", "split3_kernel")


def per_kernel(path, counter):
    tot = defaultdict(float)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            if any(e in name for e in EXCLUDE):
                continue
            tot[name.split("(")[0]] += float(r["Counter_Value"]) * 1024.0
    return tot


def main():
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    f_raw, w = sum(fetch.values()) / steps, sum(write.values()) / steps
    gn_f = sum(v for k, v in fetch.items() if "gn_apply_silu" in k)
    gn_w = sum(v for k, v in write.items() if "gn_apply_silu" in k)
    out = {
        "bytes_per_step": 2.0 * f_raw + w,
        "fetch_size_raw_bytes_per_step": f_raw,
        "write_size_bytes_per_step": w,
        "gn_apply_fetch_over_write_raw": (gn_f / gn_w) if gn_w else None,
        "by_kernel_GB_per_step": {k: round((2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) / steps / 1e9, 3)
                                  for k in sorted(set(fetch) | set(write),
                                                  key=lambda k: -(2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)))[:12]},
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (each with --kernel-trace "
                  f"only) over `bench.py --steps 3 --warmup 1 --no-graph`; counter values (KiB) summed over the "
                  f"kernels of the {steps} denoising steps (weight-packing and copy kernels excluded), x1024, /{steps}. "
                  "gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of the bytes of wide coalesced "
                  "streaming reads -> doubled. Infinity-Cache hits are counted as fetches, so this is traffic beyond "
                  "L2, not necessarily DRAM.",
    }
    dst = Path(__file__).resolve().parent / "hbm_traffic.json"
    dst.write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
