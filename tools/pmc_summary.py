#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes written by tools/pmc_run.sh into a markdown table + a small JSON file.

    python tools/pmc_summary.py gpurun_out/pmcN "<kernel name substring>" profiles/rNN_pmc_xxx

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch
(TCC_EA0_RDREQ x 64 B); on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes, so the read figure is doubled before it is
compared with a byte count.  Infinity-Cache hits are included in these fabric-side counters (they are an upper bound of HBM bytes).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_hash import trace_kernel_hash   # noqa: E402

src, pattern, out = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(list)
meta = {}
# Only the full-size launches count (the persistent grid = every resident workgroup slot); smaller grids are the chunks of the host-array leg.  The
# first full-size launch of a bench.py run traces the coherent primary rays that the bounce rays are made from: not the workload either.
for p in sorted(glob.glob(os.path.join(src, "p*", "*counter_collection.csv"))):
    rows = [r for r in csv.DictReader(open(p)) if pattern in r["Kernel_Name"]]
    if not rows:
        continue
    full = max(int(r["Grid_Size"]) for r in rows)
    seen_first = {}
    for r in rows:
        if int(r["Grid_Size"]) != full:
            continue
        if not seen_first.get(r["Counter_Name"]):
            seen_first[r["Counter_Name"]] = True           # skip the primary-ray launch
            continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta = dict(kernel=r["Kernel_Name"], grid=int(r["Grid_Size"]), workgroup=int(r["Workgroup_Size"]), lds=int(r["LDS_Block_Size"]),
                    vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), scratch=int(r["Scratch_Size"]))
avg = {k: sum(v) / len(v) for k, v in agg.items()}
n = {k: len(v) for k, v in agg.items()}
d = dict(meta=meta, counters=avg, launches=n, source_hash=trace_kernel_hash(), kernel_pattern=pattern)
if "FETCH_SIZE" in avg:
    d["hbm_read_bytes_per_launch"] = 2.0 * avg["FETCH_SIZE"] * 1024.0
if "WRITE_SIZE" in avg:
    d["hbm_write_bytes_per_launch"] = avg["WRITE_SIZE"] * 1024.0
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    d["hbm_traffic_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
if "GRBM_GUI_ACTIVE" in avg:
    d["kernel_cycles"] = avg["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
if "SQ_ACTIVE_INST_VALU" in avg and "GRBM_GUI_ACTIVE" in avg:
    d["valu_busy_frac"] = avg["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * d["kernel_cycles"])   # quad-cycles -> cycles, 1024 SIMDs
if "TCC_HIT_sum" in avg:
    d["l2_hit_rate"] = avg["TCC_HIT_sum"] / max(1.0, avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
if "SQ_WAVE_CYCLES" in avg:
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
        if k in avg:
            d[k.lower() + "_frac"] = avg[k] / avg["SQ_WAVE_CYCLES"]
json.dump(d, open(out + ".json", "w"), indent=1, sort_keys=True)
with open(out + ".md", "a") as f:
    f.write("\n| counter | per launch (mean) | launches |\n|---|---:|---:|\n")
    for k in sorted(avg):
        f.write("| %s | %.4g | %d |\n" % (k, avg[k], n[k]))
    f.write("\nDerived: " + ", ".join("%s = %.4g" % (k, v) for k, v in sorted(d.items()) if isinstance(v, float)) + "\n")
    f.write("\nKernel: `%s`\n" % json.dumps(meta))
print(json.dumps({k: v for k, v in d.items() if k not in ("counters", "launches")}, indent=1))
