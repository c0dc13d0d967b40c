#!/usr/bin/env python3
"""Files one run of profiles/collect.sh under profiles/ and writes profiles/current.json, the profile-derived
numbers bench.py quotes next to its live measurements (rocprofv3's average launch duration of the dominant
kernel, PMC traffic per launch / per step).

    python3 profiles/make_current.py gpurun_out/prof_<tag> <tag>
"""
import csv
import json
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
# candidates for the dominant kernel (bench.py picks the same way, by measured time): the fused F(2x2,3x3) kernel, and the
# bf16x3 GEMM kernel - one kernel source, template instances by loader (A as planes / as fp32) and epilogue kind
CANDIDATES = {"wino_fused_gn128_kernel": ("wino_fused_gn128_kernel",),
              "gemm_bf16x3_kernel": ("gemm_bf16x3_kernel<",)}


def main():
    src, tag = Path(sys.argv[1]), sys.argv[2]
    import gzip

    for name in ("kernel_stats.csv", "sq_summary.json", "stats_bench.json"):
        shutil.copy(src / name, HERE / f"{tag}_{name}")
    for name in ("pmc_fetch.csv", "pmc_write.csv"):   # the raw per-dispatch tables: compressed (their reduction is <tag>_hbm_traffic.json)
        with open(src / name, "rb") as fi, gzip.open(HERE / f"{tag}_{name}.gz", "wb", compresslevel=9) as fo:
            shutil.copyfileobj(fi, fo)
    out = subprocess.run([sys.executable, str(HERE / "reduce_pmc.py"), str(src / "pmc_fetch.csv"), str(src / "pmc_write.csv"), "4"],
                         capture_output=True, text=True, check=True).stdout
    traffic = json.loads(out)
    shutil.copy(HERE / "hbm_traffic.json", HERE / f"{tag}_hbm_traffic.json")
    rows = list(csv.DictReader(open(src / "kernel_stats.csv")))
    steps = 7  # collect.sh: --steps 5 --warmup 2
    member = lambda name, pats: any(p_ in name for p_ in pats)
    tot = {k: sum(float(r["TotalDurationNs"]) for r in rows if member(r["Name"], pats)) for k, pats in CANDIDATES.items()}
    name = max(tot, key=tot.get)
    pats = CANDIDATES[name]
    doms = [r for r in rows if member(r["Name"], pats)]
    calls = sum(int(r["Calls"]) for r in doms)
    launches_per_step = calls / steps
    dom_bytes = sum(v for k, v in traffic["by_kernel_GB_per_step"].items() if member(k, pats)) * 1e9
    sq = json.loads((src / "sq_summary.json").read_text())
    dom_sq = [k for k in sq["kernels"] if member(k["kernel"], pats)]
    w = sum(k["avg_us"] * k["dispatches"] for k in dom_sq)
    cur = {
        "tag": tag,
        "dominant_kernel": "kd::" + name,
        "dominant_avg_us": tot[name] / calls / 1e3,
        "dominant_launches_per_step": launches_per_step,
        "dominant_share_of_kernel_time": sum(float(r["Percentage"]) for r in doms) / 100.0,
        "dominant_bytes_per_launch": dom_bytes / launches_per_step,
        "dominant_mfma_util": sum(k["mfma_util"] * k["avg_us"] * k["dispatches"] for k in dom_sq) / w if w else None,
        "bytes_per_step": traffic["bytes_per_step"],
        "source": f"profiles/{tag}_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2 "
                  f"--no-cpu-baseline --no-kernel-classes --no-line-grid --no-cond-table`: the step without the one-off table of the time conditioning, which removes 21 small launches / 0.37 ms per step), profiles/{tag}_pmc_fetch.csv.gz + {tag}_pmc_write.csv.gz (separate "
                  "--pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md), "
                  f"profiles/{tag}_sq_summary.json (SQ_VALU_MFMA_BUSY_CYCLES pass)",
    }
    (HERE / "current.json").write_text(json.dumps(cur, indent=1) + "\n")
    print(json.dumps(cur, indent=1))


if __name__ == "__main__":
    main()
