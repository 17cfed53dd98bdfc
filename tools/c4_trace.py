"""Kernel-trace summary of tools/bench_c4.py (run ON the GPU box):
   python tools/c4_trace.py [bench_c4 args]   ->  per-kernel count / mean duration / period between back-to-back launches.
Runs rocprofv3 --kernel-trace, reads its sqlite output."""
import glob
import os
import sqlite3
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/c4trace"
subprocess.run(["rm", "-rf", out])
env = dict(os.environ, TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "-d", out, "-o", "c4", "--", sys.executable,
                os.path.join(ROOT, "tools", "bench_c4.py"), "--no-cpu-baseline", "--steps", "1"] + sys.argv[1:],
               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
db = glob.glob(out + "/**/*.db", recursive=True)[0]
rows = sqlite3.connect(db).execute("select name,start,end from kernels order by start").fetchall()
names = [r[0] for r in rows]
st = np.array([r[1] for r in rows], float)
en = np.array([r[2] for r in rows], float)
agg = {}
for i, n in enumerate(names):
    k = n.replace("void ", "").replace("gtnx::(anonymous namespace)::", "").split("(")[0][:60]
    a = agg.setdefault(k, [0, 0.0, [], None])
    a[0] += 1
    a[1] += en[i] - st[i]
    if a[3] == i - 1:
        a[2].append(st[i] - st[i - 1])
    a[3] = i
print("%-62s %7s %10s %10s %10s" % ("kernel", "calls", "total ms", "mean us", "period us"))
for k, a in sorted(agg.items(), key=lambda x: -x[1][1])[:24]:
    print("%-62s %7d %10.2f %10.2f %10.2f" % (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, (np.mean(a[2]) / 1e3) if a[2] else 0.0))
