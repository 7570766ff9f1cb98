"""Summarise rocprofv3's rocpd (sqlite) output: per-kernel launch statistics and per-kernel mean counter values.

    python tools/rocpd_summary.py <results.db> [<results.db> ...]

Durations in microseconds; counters averaged per launch (FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them).
"""
import sqlite3, sys, collections


def short(name):
    name = name.split("(")[0]
    for pre in ("void ", "ifl::"):
        name = name.replace(pre, "")
    return name[:70]


def main(paths):
    for p in paths:
        c = sqlite3.connect(p)
        names = dict(c.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
        rows = list(c.execute("select kernel_id, start, end, event_id from rocpd_kernel_dispatch"))
        per = collections.defaultdict(list)
        ev2k = {}
        for kid, s, e, ev in rows:
            per[names.get(kid, str(kid))].append((e - s) / 1000.0)
            ev2k[ev] = names.get(kid, str(kid))
        total = sum(sum(v) for v in per.values())
        print("== %s: %d dispatches, %.1f us of kernel time" % (p, len(rows), total))
        print("%-72s %6s %10s %10s %10s %6s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "%"))
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            print("%-72s %6d %10.2f %10.2f %10.2f %6.1f" % (short(k), len(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / total))
        pmc = dict(c.execute("select id, name from rocpd_info_pmc"))
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for ev, pid, val in c.execute("select event_id, pmc_id, value from rocpd_pmc_event"):
            if ev in ev2k:
                acc[ev2k[ev]][pmc.get(pid, str(pid))].append(val)
        if acc:
            print("-- counters (mean per launch)")
            for k, d in sorted(acc.items()):
                print("%-72s %s" % (short(k), "  ".join("%s=%.1f (n=%d)" % (n, sum(v) / len(v), len(v)) for n, v in sorted(d.items()))))


if __name__ == "__main__":
    main(sys.argv[1:])
