#!/usr/bin/env python3
"""r03: pwa_align_batch (fill + traceback band + walk, op lists out) over list shapes between the headline ones: which engine
takes them and what the fills and walks cost.  Run with PWA_DEBUG=1 in the environment: the library prints its own
"[pwa] overlaps/align_batch: ... fills X ms, walks Y ms" line per call on stderr; this script adds the wall time of the call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402


def main():
    pkg = bench.load_pkg()
    ctx = pkg.Context(0)
    shapes = [(16384, 64, 64), (16384, 150, 150), (4096, 300, 300), (4096, 500, 500), (2048, 1000, 1000), (512, 2000, 2000),
              (64, 10000, 10000), (8, 30000, 30000), (4096, 100, 2000), (1024, 2000, 100)]
    if len(sys.argv) > 1:   # "pairs:n:m,..."
        shapes = [tuple(int(x) for x in t.split(":")) for t in sys.argv[1].split(",")]
    for shape in shapes:
        n_pairs, n, m = shape[:3]
        embed = len(shape) > 3 and shape[3]   # "pairs:n:m:1": every text holds a mutated copy of its pattern (at a third of its length)
        pats = [bench.gen(1, 0, i, n) for i in range(n_pairs)]
        txts = [bench.gen(1, 1, i, m) for i in range(n_pairs)]
        if embed:
            for i in range(n_pairs):
                p = bytearray(pats[i])
                for x in range(7, len(p), 13):
                    p[x] = b"ACGT"[(p[x] + x) & 3]
                at = max(0, (m - n) // 3)
                txts[i] = txts[i][:at] + bytes(p) + txts[i][at + n:]
                txts[i] = txts[i][:m]
        packed = pkg.pack_sequences(pats + txts)
        pa = np.arange(n_pairs, dtype=np.uint32)
        pb = pa + np.uint32(n_pairs)
        for mode in ("nw", "sw"):
            out = ctx.align_batch_arrays(mode, packed, pa, pb, 1, -1, -1)
            t = []
            for _ in range(3):
                t0 = time.perf_counter()
                out = ctx.align_batch_arrays(mode, packed, pa, pb, 1, -1, -1, out=out)
                t.append((time.perf_counter() - t0) * 1e3)
            cells = n_pairs * n * m
            sys.stderr.flush()
            print("== %6d pairs %6d x %6d%s %s: call %.2f ms wall (%.0f GCUPS incl. copies), checksum %d" %
                  (n_pairs, n, m, " (pattern inside)" if embed else "", mode, min(t), cells / min(t) / 1e6, int(out["scores"][:n_pairs].astype(np.int64).sum())), flush=True)


if __name__ == "__main__":
    main()
