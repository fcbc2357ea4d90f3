#!/usr/bin/env python3
"""r03: a scores pass WITH end cells (pwa_scores(..., end_i_out, end_j_out): where a traceback would start) over many short patterns:
device time of pwa_batch_run on the band-less mini-stripe kernels (default) against the stripe engine (PWA_TB_ENGINE=0), and that
both return the same scores and cells."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402


def main():
    pkg = bench.load_pkg()
    shapes = [("65536 pairs 150 x 10k (1024 patterns x 64 texts)", 1024, 64, 150, 10000),
              ("262144 pairs 150 x 1000 (1024 x 256)", 1024, 256, 150, 1000),
              ("16384 pairs 250 x 5000 (256 x 64)", 256, 64, 250, 5000)]
    print("%-50s %-4s %14s %14s" % ("shape", "mode", "mini ms", "stripes ms"))
    for name, npat, ntxt, n, m in shapes:
        pats = [bench.gen(1, 0, i, n) for i in range(npat)]
        txts = [bench.gen(1, 1, i, m) for i in range(ntxt)]
        seqs = pats + txts
        pa = np.repeat(np.arange(npat, dtype=np.uint32), ntxt)
        pb = (npat + np.tile(np.arange(ntxt, dtype=np.uint32), npat)).astype(np.uint32)
        for mode in ("sw", "nw"):
            row, res = [], []
            for eng in ("auto", "0"):
                if eng == "0":
                    os.environ["PWA_TB_ENGINE"] = "0"
                ctx = pkg.Context(0)
                os.environ.pop("PWA_TB_ENGINE", None)
                b = ctx.batch(mode, seqs, pa, pb, 1, -1, -1, True)
                b.run()
                b.last_ms()
                t = []
                for _ in range(3):
                    b.run()
                    t.append(b.last_ms())
                row.append(min(t))
                res.append(b.fetch(numpy_out=True))
                kern = b.info()["kernel"]
                b.close()
                ctx.close()
            same = all(np.array_equal(x, y) for x, y in zip(res[0], res[1]))
            print("%-50s %-4s %14.3f %14.3f  %s  %s" % (name, mode, row[0], row[1], "same results" if same else "RESULTS DIFFER", kern), flush=True)
            if not same:
                sys.exit(1)


if __name__ == "__main__":
    main()
