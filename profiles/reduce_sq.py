#!/usr/bin/env python3
"""Reduces one `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE` pass over

    python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-classes

to a per-kernel table: share of the device time, matrix-pipe utilisation and where the waves' cycles go.

    python3 profiles/reduce_sq.py <counter_collection.csv> <kernel_trace.csv> > profiles/r01_v8_sq_summary.json

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs) / 1024 SIMDs (the counter sums the busy
cycles of every SIMD's matrix pipe; GRBM_GUI_ACTIVE sums the 8 XCDs).  clock_ghz = GRBM_GUI_ACTIVE / 8 /
duration (MI355X_MICROARCH.md, DVFS: reads high on dispatches well under 0.3 ms).  wait_any = parked on
s_waitcnt / barrier, wait_inst = issue stall (matrix pipe busy / dependency), active = issuing: fractions of
SQ_WAVE_CYCLES."""
import collections
import csv
import json
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(sys.argv[2]))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"])
            t = trace[r["Dispatch_Id"]]
            agg[k]["dur_ns"] += float(t["End_Timestamp"]) - float(t["Start_Timestamp"])
            agg[k]["n"] += 1
    total = sum(v["dur_ns"] for v in agg.values())
    out = []
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["dur_ns"]):
        if v["dur_ns"] < 0.003 * total:
            continue
        gui = v["GRBM_GUI_ACTIVE"] / 8.0
        wc = max(v["SQ_WAVE_CYCLES"], 1.0)
        out.append({"kernel": k, "dispatches": int(v["n"]), "share_of_device_time": round(v["dur_ns"] / total, 4),
                    "avg_us": round(v["dur_ns"] / v["n"] / 1e3, 1), "clock_ghz": round(gui / v["dur_ns"], 2),
                    "mfma_util": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / gui / 1024.0, 3) if gui else 0.0,
                    "wait_any": round(v["SQ_WAIT_ANY"] / wc, 2), "wait_inst": round(v["SQ_WAIT_INST_ANY"] / wc, 2),
                    "active": round(v["SQ_ACTIVE_INST_ANY"] / wc, 2)})
    json.dump({"device_ms_in_trace": round(total / 1e6, 2), "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
