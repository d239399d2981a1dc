#!/usr/bin/env python3
"""tools/region_count.py <kernel.s>: VALU / SALU / LDS / VMEM instruction counts per basic block of a kernel's assembly (blocks >= 8 VALU)."""
import re, sys
cur, out = "entry", {}
order = []
for ln in open(sys.argv[1]):
    m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):", ln)
    if m:
        cur = m.group(1) or ("bb." + m.group(2)); continue
    t = ln.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    c = out.setdefault(cur, [0, 0, 0, 0]);
    if cur not in order: order.append(cur)
    if t.startswith("v_"): c[0] += 1
    elif t.startswith("s_"): c[1] += 1
    elif t.startswith("ds_"): c[2] += 1
    elif re.match(r"(global|flat|buffer|scratch)_", t): c[3] += 1
tot = [0, 0, 0, 0]
for k in order:
    c = out[k]
    for i in range(4): tot[i] += c[i]
    if c[0] >= int(sys.argv[2]) if len(sys.argv) > 2 else c[0] >= 8: print("%-14s valu %4d salu %4d lds %3d vmem %3d" % (k, *c))
print("total          valu %4d salu %4d lds %3d vmem %3d" % tuple(tot))
