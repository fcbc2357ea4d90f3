#!/usr/bin/env python3
"""Does the first launch after a host-side pause run slower on the device?  C3 batch: device time (HIP events) of back-to-back
runs and of runs that follow an idle gap of 5 / 20 / 50 / 200 ms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = bench.load_pkg()
ctx = pkg.Context(0)
mode, seqs, pa, pb, scoring, desc = bench.build_c3(0)
b = ctx.batch(mode, seqs, pa, pb, *scoring)
for i in range(3):
    b.run(); print("back-to-back run %d: %.2f ms" % (i, b.last_ms()))
for gap in (0.005, 0.02, 0.05, 0.2, 0.0):
    time.sleep(gap)
    b.run(); print("after %3.0f ms idle: %.2f ms" % (gap * 1e3, b.last_ms()))
import numpy as np
scores = np.zeros(len(pa), dtype=np.int32)
packed = pkg.pack_sequences(seqs)
for rep in range(3):
    b.close()
    t0 = time.perf_counter(); b = ctx.batch(mode, packed, pa, pb, *scoring); t1 = time.perf_counter()
    b.run(); t2 = time.perf_counter(); ms = b.last_ms(); t3 = time.perf_counter(); b.fetch_into(scores); t4 = time.perf_counter()
    print("fresh batch %d: create %.1f ms, run() call %.2f ms, wait-for-device %.1f ms (device %.2f ms), fetch %.2f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, ms, (t4 - t3) * 1e3))
keep = []
for rep in range(3):
    keep.append(b)   # previous batches stay alive: no hipFree between create and run
    t0 = time.perf_counter(); b = ctx.batch(mode, packed, pa, pb, *scoring); t1 = time.perf_counter()
    b.run(); t2 = time.perf_counter(); ms = b.last_ms(); t3 = time.perf_counter()
    print("fresh batch, nothing freed %d: create %.1f ms, wait-for-device %.1f ms (device %.2f ms)" % (rep, (t1 - t0) * 1e3, (t3 - t2) * 1e3, ms))
    b.run(); t4 = time.perf_counter(); ms = b.last_ms(); t5 = time.perf_counter()
    print("   second run of it: wait-for-device %.1f ms (device %.2f ms)" % ((t5 - t4) * 1e3, ms))
for gap in (0.0, 0.001, 0.003, 0.005, 0.01, 0.02, 0.05):
    time.sleep(gap)
    t0 = time.perf_counter(); b.run(); ms = b.last_ms(); t1 = time.perf_counter()
    print("same batch after %4.0f ms of host sleep: wall %.1f ms, device %.2f ms" % (gap * 1e3, (t1 - t0) * 1e3, ms))
