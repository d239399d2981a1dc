"""Per-kernel summary of a rocprofv3 --kernel-trace run (reads the rocpd sqlite database): python tools/kstats.py <dir>"""
import glob, sqlite3, subprocess, sys
for f in glob.glob(sys.argv[1] + '/**/*.db', recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    q = f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows)
    print("| kernel | calls | total us | avg us | min us | % |\n|---|---:|---:|---:|---:|---:|")
    for r in rows:
        name = r[0]
        try:
            name = subprocess.run(["c++filt", name.replace(".kd", "")], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            pass
        name = name.replace("(anonymous namespace)::", "").split("(")[0]
        print("| %s | %d | %.1f | %.1f | %.1f | %.2f |" % (name, r[1], r[2], r[3], r[4], 100 * r[2] / tot))
    # The traversal kernel is launched at many sizes in one bench run (one-ray calls, chunks of the host-array path, the timed batches): its row above mixes them.
    # What bench.py's HIP events time are the full-size launches of the timed region -- the longest run of back-to-back, non-overlapping full-grid launches on one queue.
    trows = list(db.execute(f"select d.start, d.end, d.grid_size_x, d.queue_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%trace_kernel_q%' order by d.start"))
    if trows:
        import collections
        top = collections.Counter(r[4] for r in trows).most_common(1)[0][0]
        trows = [r for r in trows if r[4] == top]
        gmax = max(r[2] for r in trows)
        print("\nThe traversal kernel by launch size (grid threads: calls, average us): " + ", ".join("%d: %d, %.1f" % (g, len(v), sum(v) / len(v) / 1e3) for g, v in
              sorted(collections.defaultdict(list, {g: [r[1] - r[0] for r in trows if r[2] == g] for g in set(r[2] for r in trows)}).items())))
        full = [r for r in trows if r[2] == gmax]
        lone = [r for i, r in enumerate(full) if not any(o is not r and o[0] < r[1] and o[1] > r[0] for o in full[max(0, i - 6):i + 7])]
        runs, cur = [], lone[:1]
        for a, b in zip(lone, lone[1:]):
            if b[3] == a[3] and b[0] - a[1] < 50000: cur.append(b)
            else: runs.append(cur); cur = [b]
        runs.append(cur)
        best = max(runs, key=len)
        dd = [(x[1] - x[0]) / 1e3 for x in best]
        print("Full-grid launches: %d, of them not overlapping another one: %d (average %.1f us).  **The timed region = the longest run of back-to-back lone launches on one queue: %d launches, "
              "average %.1f us, min %.1f us** -- the figure bench.py's HIP events report as `roofline.kernel_ms_avg_overlapping`." % (len(full), len(lone), sum((x[1] - x[0]) / 1e3 for x in lone) / max(1, len(lone)), len(best), sum(dd) / len(dd), min(dd)))
