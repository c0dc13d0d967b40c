"""Scans the gfx950 assembly of the engine's kernels for vector-memory loads whose value is waited for at once
(`s_waitcnt vmcnt(0)` within a few instructions of the load, nothing issued in between): the pattern behind the serialised
input-transform loads, the fp32-A loader waves and the 16-byte epilogue of round 5 (profiles/README.md).
  python scratch/isa_scan/scan_waits.py /tmp/scan_*.s"""
import re, sys
for path in sys.argv[1:]:
    s = open(path).read()
    for m in re.finditer(r"^(_ZN\S+):[^\n]*\n(.*?)\n\.Lfunc_end", s, re.S | re.M):
        name, body = m.group(1), m.group(2).split('\n')
        ins = [l.split(';')[0].strip() for l in body]
        ins = [l for l in ins if l and not l.startswith('.') and not l.endswith(':')]
        loads = [i for i, l in enumerate(ins) if re.match(r'(buffer|global|flat)_load', l) and ' lds' not in l]
        hot = 0
        for i in loads:
            for j in range(i + 1, min(i + 5, len(ins))):
                if re.match(r'(buffer|global|flat)_(load|store)', ins[j]):
                    break
                if 's_waitcnt' in ins[j] and 'vmcnt(0)' in ins[j]:
                    hot += 1
                    break
        if loads and hot >= 4:
            print(f"{path.split('scan_')[-1]:28s} {name[:70]:70s} loads {len(loads):4d}  waited-at-once {hot}")
