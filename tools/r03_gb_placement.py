#!/usr/bin/env python3
"""r03: does the `gb` fill depend on where its bands land?  One process, six contexts one after the other (each allocates its own 6.6 GB
code band and 26 GB score band, and frees them when it closes): fill time of the 4096 x (150 x 10k) SW batch in each, and the device
addresses the allocations got are not visible from here -- only whether the time follows the context or the process."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402


def main():
    pkg = bench.load_pkg()
    n_pairs, n, m = 4096, 150, 10000
    pats = [bench.gen(1, 0, i, n) for i in range(n_pairs)]
    txts = [bench.gen(1, 1, i % 256, m) for i in range(n_pairs)]
    packed = pkg.pack_sequences(pats + txts)
    pa = np.arange(n_pairs, dtype=np.uint32)
    pb = pa + np.uint32(n_pairs)
    for trial in range(6):
        ctx = pkg.Context(0)
        ctx.set_score_band(True)
        out = ctx.align_batch_arrays("sw", packed, pa, pb, 1, -1, -1)
        t = []
        for _ in range(5):
            out = ctx.align_batch_arrays("sw", packed, pa, pb, 1, -1, -1, out=out)
            t.append(ctx.align_stats()["fill_ms"])
        print("context %d: fill ms %s" % (trial, " ".join("%.3f" % x for x in t)), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
