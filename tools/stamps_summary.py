#!/usr/bin/env python3
"""Condense a PWA_STAMPS file (+ its .trace) into the few lines kept under profiles/."""
import sys
import numpy as np
path = sys.argv[1]
d = []
for line in open(path):
    a = [int(x) for x in line.split()]
    d.append([x if x < 2**63 else x - 2**64 for x in a])
d = np.array(d)
dur = (d[:, 4] - d[:, 2]) / 100.0
lag = np.diff(d[:, 2]) / 100.0
print("stripes %d; s_memrealtime ticks of 10 ns; columns of the raw file: stripe, start, start of chunk 1, start of chunk 5, end" % len(d))
print("duration (chunk 1 -> end) of the first stripe of every 4-stripe workgroup, us:")
print(" ", np.round(dur[::4], 0).astype(int).tolist())
intra = [lag[i] for i in range(len(lag)) if (i + 1) % 4 != 0]
inter = [lag[i] for i in range(len(lag)) if (i + 1) % 4 == 0]
print("start lag between consecutive stripes: inside a workgroup mean %.1f us (min %.1f max %.1f), across workgroups mean %.1f us (min %.1f max %.1f)"
      % (np.mean(intra), np.min(intra), np.max(intra), np.mean(inter), np.min(inter), np.max(inter)))
print("last stripe starts at %.0f us and ends at %.0f us" % (d[-1, 2] / 100.0, d[-1, 4] / 100.0))
try:
    t = np.loadtxt(path + ".trace").astype(np.int64)
    for k in range(1, 5):
        x = t[:, k]
        x = x[(x > 0) & (x < 10**9)]
        if len(x) < 10:
            continue
        dt = np.diff(x)
        print("traced stripe +%d: %d chunks of 16 steps, chunk time in ticks: median %.0f mean %.1f p90 %.0f p99 %.0f; histogram in bins of 20 ticks from 0: %s"
              % (k - 1, len(x), np.median(dt), dt[600:].mean(), np.percentile(dt, 90), np.percentile(dt, 99), np.bincount(np.clip(dt, 0, 400) // 20)[:16].tolist()))
except Exception as e:   # noqa: BLE001
    print("(no trace: %s)" % e)
