"""Summarises per-op CSVs of scratch/dump_ops.py (kd_unet_profile): total, per class, fused-conv layers by Cin.
   python scratch/ops_summary.py a.csv [b.csv ...]"""
import collections, csv, re, sys

def load(path):
    rows = [r for r in csv.reader(open(path)) if len(r) == 5 and r[0].isdigit()]
    return [(r[1], int(r[2]), float(r[3]), int(r[4])) for r in rows]

for path in sys.argv[1:]:
    rows = load(path)
    tot = sum(r[2] for r in rows)
    cls = collections.defaultdict(lambda: [0, 0.0, 0])
    for l, m, us, mf in rows:
        k = re.sub(r"\d+", "#", l)
        if l.startswith("wino fused"):
            k = "wino fused Cin" + re.search(r"Cin(\d+)", l).group(1)
        c = cls[k]
        c[0] += 1; c[1] += us; c[2] += mf
    print(f"== {path}: {len(rows)} launches, {tot / 1e3:.3f} ms")
    fused = sum(v[1] for k, v in cls.items() if k.startswith("wino fused"))
    fmf = sum(v[2] for k, v in cls.items() if k.startswith("wino fused"))
    if fused:
        print(f"   fused Winograd convs: {fused / 1e3:.3f} ms, {2 * fmf / fused / 1e6 / 157.3:.3f} of peak")
    for k, (n, us, mf) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
        frac = f"{2 * mf / us / 1e6 / 157.3:.3f}" if mf else "     "
        print(f"   {us:9.1f} us {n:4d} x {us / n:8.1f}  {frac}  {k}")
