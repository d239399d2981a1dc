#!/usr/bin/env python3
"""HBM-side traffic of a whole commit from the rocprofv3 --pmc passes of tools/pmc_run.sh over tests/gpu_build_only.py: FETCH_SIZE x 2 (gfx950: 128-byte requests count
as 64) + WRITE_SIZE, in KiB per dispatch, summed per kernel and divided by the number of commits in the run (= dispatches of small_build).
    python tools/commit_traffic.py gpurun_out/r06v/pmc_build profiles/r06_commit_traffic   ->  .md + .json (bench.py puts the total beside its formula in build.roofline)"""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(float))
ncommit = 0
for p in sorted(glob.glob(os.path.join(src, "p*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(p)):
        c = r["Counter_Name"]
        if c not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        per[k][c] += float(r["Counter_Value"])
        if c == "FETCH_SIZE" and k == "small_build":
            ncommit += 1
rows = sorted(((v["FETCH_SIZE"] * 2048.0 / ncommit, v["WRITE_SIZE"] * 1024.0 / ncommit, k) for k, v in per.items()), key=lambda t: -(t[0] + t[1]))
total_r, total_w = sum(r[0] for r in rows), sum(r[1] for r in rows)
md = ["| kernel | read MB | written MB |", "|---|---:|---:|"] + ["| %s | %.1f | %.1f |" % (k, r / 1e6, w / 1e6) for r, w, k in rows if r + w > 1e6]
md.append("| **commit** | **%.1f** | **%.1f** |" % (total_r / 1e6, total_w / 1e6))
open(out + ".md", "w").write("HBM-side traffic per commit (%d commits in the run; FETCH_SIZE x 2 + WRITE_SIZE, Infinity-Cache hits included: an upper bound of HBM bytes)\n\n" % ncommit + "\n".join(md) + "\n")
json.dump({"commits": ncommit, "read_bytes_per_commit": total_r, "write_bytes_per_commit": total_w, "traffic_bytes_per_commit": total_r + total_w,
           "per_kernel": {k: {"read": r, "write": w} for r, w, k in rows}}, open(out + ".json", "w"), indent=1)
print("\n".join(md[-12:] if len(md) > 12 else md)); print("commits", ncommit)
