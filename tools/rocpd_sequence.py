"""Kernel dispatches of a rocprofv3 rocpd database in time order: name, duration, gap to the previous kernel's end.
    python tools/rocpd_sequence.py <results.db> [first] [count]"""
import sqlite3, sys
from rocpd_summary import short

c = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
names = dict(c.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
rows = sorted(c.execute("select start, end, kernel_id from rocpd_kernel_dispatch"))
prev = None
for i, (s, e, k) in enumerate(rows):
    if first <= i < first + count:
        print("%5d %-60s %8.2f us   gap %7.2f us" % (i, short(names.get(k, str(k)))[:60], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
