#!/usr/bin/env python3
"""Aggregate the samples of GTN_HOST_SAMPLE (tools/nullhip/host_step.cpp): self time by function
and inclusive time by function, resolved with nm.   usage: report.py <samples> [top]"""
import bisect
import collections
import subprocess
import sys

maps, samples, base = [], [], {}
for ln in open(sys.argv[1]):
    if ln.startswith("M "):
        f = ln[2:].split()
        if len(f) >= 6:
            lo, hi = (int(x, 16) for x in f[0].split("-"))
            base[f[5]] = min(base.get(f[5], lo), lo)  # first PT_LOAD of a shared object sits at vaddr 0
            if "x" in f[1]:
                maps.append((lo, hi, f[5]))
    elif ln.startswith("S"):
        samples.append([int(x, 16) for x in ln.split()[1:]])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
syms = {}


def table(path):
    if path not in syms:
        out = subprocess.run(["nm", "-C", "--defined-only", path], capture_output=True, text=True).stdout
        out += subprocess.run(["nm", "-C", "-D", "--defined-only", path], capture_output=True, text=True).stdout
        t = sorted({(int(p[0], 16), " ".join(p[2:])) for p in (l.split() for l in out.splitlines())
                    if len(p) >= 3 and p[1] in "tTwW"})
        syms[path] = ([a for a, _ in t], [n for _, n in t])
    return syms[path]


def resolve(pc):
    for lo, hi, path in maps:
        if lo <= pc < hi:
            addrs, names = table(path)
            rel = pc - base[path]
            i = bisect.bisect_right(addrs, rel) - 1
            return (names[i] if i >= 0 else "?") + " [" + path.split("/")[-1] + "]"
    return "?"


self_t, incl = collections.Counter(), collections.Counter()
for st in samples:
    fr = [resolve(pc) for pc in st[2:]]  # drop the handler and the signal trampoline
    if not fr:
        continue
    self_t[fr[0]] += 1
    for name in set(fr):
        incl[name] += 1
n = len(samples)
print(f"{n} samples (100 us of main-thread CPU each)")
print("--- self")
for k, v in self_t.most_common(top):
    print(f"{100.0 * v / n:6.2f}%  {k[:150]}")
print("--- inclusive")
for k, v in incl.most_common(top):
    print(f"{100.0 * v / n:6.2f}%  {k[:150]}")

# who pays for the allocator: nearest frame outside libc / libstdc++ above a malloc / free sample
alloc = collections.Counter()
for st in samples:
    fr = [resolve(pc) for pc in st[2:]]
    if not fr or not any(k in fr[0] for k in ("malloc", "free", "morecore", "operator new", "operator delete")):
        continue
    for name in fr:
        if "libc.so" not in name and "libstdc++" not in name:
            alloc[name] += 1
            break
print("--- allocator time by caller")
for k, v in alloc.most_common(top):
    print(f"{100.0 * v / n:6.2f}%  {k[:150]}")
