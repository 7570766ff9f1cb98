"""The largest idle periods between consecutive kernels of a rocprofv3 rocpd database, and what the host was doing meanwhile.
    python tools/stall_gaps.py <results.db> [n]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
def tab(prefix):
    m = [t for t in tabs if t.startswith(prefix)]
    return m[0] if m else None
kd = tab("rocpd_kernel_dispatch"); ks = tab("rocpd_info_kernel_symbol")
names = dict(c.execute("select id, kernel_name from %s" % ks))
rows = sorted(c.execute("select start, end, kernel_id from %s" % kd))
t0 = rows[0][0]
gaps = sorted(((rows[i + 1][0] - rows[i][1], i) for i in range(len(rows) - 1)), reverse=True)[:n]
dur = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print("%d kernels, %.1f ms of kernel time in a span of %.1f ms; the %d largest gaps:" % (len(rows), dur / 1e6, span / 1e6, n))
reg = tab("rocpd_region"); st = tab("rocpd_string")
api = []
if reg and st:
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % reg)]
    strs = dict(c.execute("select id, string from %s" % st))
    if "name_id" in cols:
        api = sorted((s, e, strs.get(nm, str(nm))) for s, e, nm in c.execute("select start, end, name_id from %s" % reg))
for g, i in gaps:
    s, e = rows[i][1], rows[i + 1][0]
    print("  gap %9.1f us after kernel #%d (%s) at t = %.1f ms" % (g / 1e3, i, names.get(rows[i][2], "?")[:48], (s - t0) / 1e6))
    inside = [(a, b, nm) for a, b, nm in api if b > s and a < e]
    from collections import Counter
    cnt = Counter(nm for _, _, nm in inside)
    longest = sorted(inside, key=lambda r: r[0] - r[1])[:3]
    print("      host API calls overlapping it: %s" % ", ".join("%s x%d" % kv for kv in cnt.most_common(6)))
    for a, b, nm in longest:
        print("      longest: %-32s %9.1f us (starts %+.1f us relative to the gap's start)" % (nm[:32], (b - a) / 1e3, (a - s) / 1e3))
