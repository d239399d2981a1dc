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
