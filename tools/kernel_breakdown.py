"""rocprofv3 kernel trace (csv) -> share of device time per kernel.  With a window (ms) only the dispatches of the last
<window> before the final kernel's end are counted: the steady replays of a run whose beginning is warm-up / solver search.
usage: kernel_breakdown.py DIR [rows] [tail_ms]"""
import collections, csv, glob, sys
rows_n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
tail_ms = float(sys.argv[3]) if len(sys.argv) > 3 else None
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit("no kernel_trace.csv under " + sys.argv[1])
rows = list(csv.DictReader(open(files[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
end = max(e[1] for e in ev)
if tail_ms is not None:
    ev = [e for e in ev if e[0] >= end - tail_ms * 1e6]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    agg[n][0] += e - s
    agg[n][1] += 1
tot = sum(v[0] for v in agg.values())
span = (end - min(e[0] for e in ev)) / 1e6
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:rows_n]:
    print("%6.2f%% %7d calls %9.1f us avg  %s" % (100 * t / tot, c, t / c / 1e3, n[:150]))
print("busy %.2f ms of a %.2f ms window, %d dispatches, %d kernels" % (tot / 1e6, span, len(ev), len(agg)))
