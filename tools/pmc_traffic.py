#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_run.sh into profiles/<name>.json:
per-kernel HBM-side bytes per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM
section) prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per
128 B request of a wide coalesced read, i.e. reads half the true bytes -> doubled.  The doubling was
checked here on encode_gather_kernel (a pure 16 B/lane copy of C bytes: FETCH_SIZE*2 = C within 2 %,
WRITE_SIZE = C); for the gather-dominated kernels (far-table probes) it is an upper bound.
Usage: tools/pmc_traffic.py fetch.db write.db out.json workload_bytes [workload name, default enwik] [commit the library was built from]"""
import json
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    acc = defaultdict(list)
    for k, c, v in cur.execute(f"select {kn}, counter_name, value from counters_collection"):
        if c == counter:
            acc[k.split("(")[0].replace("void ", "")].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"workload_bytes": int(sys.argv[4]), "workload": sys.argv[5] if len(sys.argv) > 5 else "enwik", "unit": "bytes per launch",
       "correction": "FETCH_SIZE KiB * 1024 * 2 (gfx950 half-count of wide reads) + WRITE_SIZE KiB * 1024",
       "commit": sys.argv[6] if len(sys.argv) > 6 else "not recorded", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if "mlz::" not in k:
        continue
    f, w = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    out["kernels"][k.replace("mlz::", "")] = {"fetch_raw": int(f), "write_raw": int(w), "traffic": int(2 * f + w)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
