#!/usr/bin/env python3
"""Where does a single pair's fill time go?  fill_ms of NW pairs n x m for a ladder of n at fixed m and of m at fixed n
(PWA_FORCE_RL / PWA_FORCE_W select the geometry): the slope in m is the cost of one anti-diagonal step of a stripe, the
slope in n the lag one more stripe / one more workgroup hand-off adds."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = bench.load_pkg()
ctx = pkg.Context(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "nw"


def fill(n, m, reps=3):
    p, t = bench.gen(1, 0, 0, n), bench.gen(1, 1, 0, m)
    best = 1e9
    for _ in range(reps):
        ctx.align(mode, p, t, 1, -1, -1, raw=True)
        best = min(best, ctx.align_stats()["fill_ms"])
    return best


print("mode", mode, "RL", os.environ.get("PWA_FORCE_RL"), "W", os.environ.get("PWA_FORCE_W"))
for m in (20000, 100000):
    for n in (64, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
        ms = fill(n, m)
        print("n %6d m %6d fill %8.3f ms  = %7.1f cycles per column (2.4 GHz)" % (n, m, ms, ms * 1e-3 * 2.4e9 / m), flush=True)
