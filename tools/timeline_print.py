"""Prints the last N ms of a rocprofv3 csv kernel + memory-copy trace as one timeline (start, duration, what)."""
import csv, glob, sys
d = sys.argv[1]; span_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s" % r.get("Direction", r.get("Name", "copy"))))
ev.sort()
t_end = max(e[1] for e in ev)
ev = [e for e in ev if e[0] > t_end - span_ms * 1e6]
t0 = ev[0][0]
for s, e, n in ev:
    if (e - s) > 20000 or n.startswith("C"):
        print("%9.3f ms  +%8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
