"""Reads a rocprofv3 kernel-trace CSV of scratch/w4f_ab.py: durations of the two fused conv kernels, in launch order."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows
       if "wino_fused_gn128_kernel" in r["Kernel_Name"] or "wino4_fused_gn_kernel" in r["Kernel_Name"]]
# launches alternate F(2x2), F(4x4), three rounds per shape
per = defaultdict(list)
shape = 0
for n in range(0, len(seq), 6):
    grp = seq[n:n + 6]
    a = [d for k, d in grp if "gn128" in k]
    b = [d for k, d in grp if "wino4_fused" in k]
    print(f"shape {shape}: F(2x2) fused {min(a):9.1f} us   F(4x4) fused {min(b):9.1f} us   ratio {min(b) / min(a):.3f}")
    shape += 1
