"""Timeline of one step from a rocprofv3 --kernel-trace CSV: kernels in start order with the idle time in front of
each (diagnostic).   usage: python tools/trace_gaps.py <kernel_trace.csv> <anchor kernel substring> [step index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
idx = [i for i, e in enumerate(ev) if anchor in e[2]]
if len(idx) < 3:
    sys.exit("anchor kernel not found often enough")
a, b = idx[which], idx[which + 1]
t0 = ev[a][0]
busy = 0
prev_end = ev[a][0]
print("step from %s to the next one: %.1f us" % (anchor, (ev[b][0] - t0) / 1e3))
for s, e, n in ev[a:b]:
    gap = (s - prev_end) / 1e3
    busy += (e - s)
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, n[:110]))
    prev_end = max(prev_end, e)
print("busy %.1f us of %.1f" % (busy / 1e3, (ev[b][0] - t0) / 1e3))
