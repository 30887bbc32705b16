#!/usr/bin/env python3
"""Per-kernel PMC averages from rocprofv3 rocpd sqlite outputs.
Usage: python tools/rocpd_pmc.py db1 [db2 ...]"""
import sqlite3
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
    # expected columns: kernel name, counter name, value, dispatch id
    kn = [c for c in cols if c in ("kernel_name", "name")][0] if any(c in ("kernel_name", "name") for c in cols) else None
    q = "select * from counters_collection limit 1"
    if kn is None:
        print(path, cols)
        continue
    cn = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    vn = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    did = "dispatch_id" if "dispatch_id" in cols else cols[0]
    acc = defaultdict(lambda: defaultdict(list))
    for k, c, v, d in cur.execute(f"select {kn}, {cn}, {vn}, {did} from counters_collection"):
        acc[k][c].append(v)
    print("#", path)
    for k in sorted(acc):
        if not k.startswith("mlz::") and "mlz::" not in k:
            continue
        items = ["%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())]
        print("%-48s n=%d  %s" % (k[:48], len(next(iter(acc[k].values()))), "  ".join(items)))
