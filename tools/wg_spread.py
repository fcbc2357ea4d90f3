#!/usr/bin/env python3
"""Spread of the intrinsic speed of independent workgroups: a batch of 96 NW pairs 1024 x 100000 (one 4-stripe workgroup each,
no hand-off between them), per-stripe time stamps (PWA_STAMPS)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
os.environ["PWA_STAMPS"] = out = os.path.join(ROOT, "gpurun_out", "r02", "stamps_spread.txt")
pkg = bench.load_pkg()
ctx = pkg.Context(0)
npairs, n, m = 96, int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 100000
seqs = [bench.gen(1, 0, i, n) for i in range(npairs)] + [bench.gen(1, 1, 0, m)]
pa = list(range(npairs))
pb = [npairs] * npairs
ctx.overlaps("nw", seqs, pa, pb, 1, -1, -1)
ctx.overlaps("nw", seqs, pa, pb, 1, -1, -1)
print(ctx.align_stats())
d = []
for line in open(out):
    a = [int(x) for x in line.split()]
    d.append([x if x < 2**63 else x - 2**64 for x in a])
d = np.array(d)
dur = (d[:, 4] - d[:, 2]) / 100.0
per = len(d) // npairs
print("stripes per pair", per)
for w in range(per):
    x = dur[w::per]
    print("stripe %d of each pair: duration us min %.0f median %.0f max %.0f" % (w, x.min(), np.median(x), x.max()))
print("sorted durations of stripe 0:", np.sort(dur[0::per]).round(0).astype(int).tolist())
