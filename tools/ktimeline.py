"""Launch-by-launch timeline of the LAST commit in a rocprofv3 --kernel-trace run (rocpd sqlite): python tools/ktimeline.py <dir>
Prints every launch from the last build_begin on: start offset, duration, idle gap before it (all us), and per-kernel sums of duration and of gaps."""
import glob, sqlite3, subprocess, sys, collections
def short(name):
    try:
        name = subprocess.run(["c++filt", name.replace(".kd", "")], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    return name.replace("(anonymous namespace)::", "").split("(")[0]
for f in glob.glob(sys.argv[1] + '/**/*.db', recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = list(db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    names = {}
    rows = [(names.setdefault(r[0], short(r[0])), r[1], r[2]) for r in rows]
    last = max(i for i, r in enumerate(rows) if r[0] == 'build_begin')
    seq = rows[last:]
    # the commit ends with the last tri_records / copy before the next non-build kernel
    end = len(seq)
    for i, r in enumerate(seq):
        if r[0].startswith('void trace') or r[0].startswith('trace_'):
            end = i; break
    seq = seq[:end]
    t0 = seq[0][1]; prev = t0
    dur = collections.Counter(); gap = collections.Counter(); cnt = collections.Counter()
    verbose = len(sys.argv) > 2
    for n, s, e in seq:
        if verbose: print("%9.1f  %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n))
        dur[n] += (e - s) / 1e3; gap[n] += max(0, s - prev) / 1e3; cnt[n] += 1; prev = max(prev, e)
    print("commit: %.1f us from the first launch to the end of the last, %d launches, busy %.1f us, idle %.1f us" % ((prev - t0) / 1e3, len(seq), sum(dur.values()), sum(gap.values())))
    print("| kernel | launches | busy us | idle before us |\n|---|---:|---:|---:|")
    for n, v in dur.most_common():
        print("| %s | %d | %.1f | %.1f |" % (n, cnt[n], v, gap[n]))
