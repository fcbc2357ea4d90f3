#!/usr/bin/env python3
"""r03: what a scores pass costs on each engine for lists of few long pairs -- and, after the
work-aware routing went in, what the library picks by itself.  Device time of pwa_batch_run per shape:
    default            whatever pwa_batch_create routes to
    want_end=1         the pair engine (anti-diagonal stripes, no band): exact end cells force it
Also times `hw2_amd -l` on the C2 pair (wall) and checks every score against the oracle's score-only form."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench   # noqa: E402
import oracle_lib as O   # noqa: E402


def shapes():
    out = []
    for k in (1, 3, 8, 64):
        out.append(("%d pairs 10k x 10k, own texts" % k, [bench.gen(1, 0, i, 10000) for i in range(k)],
                    [bench.gen(1, 1, i, 10000) for i in range(k)], None))
    out.append(("64 pairs 10k x 10k, ONE text", [bench.gen(1, 0, i, 10000) for i in range(64)], [bench.gen(1, 1, 0, 10000)], "shared"))
    out.append(("512 pairs 2000 x 2000, own texts", [bench.gen(1, 0, i, 2000) for i in range(512)],
                [bench.gen(1, 1, i, 2000) for i in range(512)], None))
    out.append(("4096 pairs 150 x 10k, 256 texts", [bench.gen(1, 0, i, 150) for i in range(4096)],
                [bench.gen(1, 1, i, 10000) for i in range(256)], "mod"))
    out.append(("1 long (10k x 10k) + 1000 short (150 x 2000)", [bench.gen(1, 0, 0, 10000)] + [bench.gen(1, 0, 1 + i, 150) for i in range(1000)],
                [bench.gen(1, 1, 0, 10000)] + [bench.gen(1, 1, 1 + i, 2000) for i in range(1000)], None))
    return out


def main():
    pkg = bench.load_pkg()
    ctx = pkg.Context(0)
    print("%-52s %-6s %12s %12s  %s" % ("shape", "mode", "default ms", "pair-eng ms", "kernel picked"))
    for name, pats, txts, how in shapes():
        seqs = pats + txts
        n = len(pats)
        pa = np.arange(n, dtype=np.uint32)
        if how == "shared":
            pb = np.full(n, n, dtype=np.uint32)
        elif how == "mod":
            pb = (n + np.arange(n) % len(txts)).astype(np.uint32)
        else:
            pb = (n + np.arange(n)).astype(np.uint32)
        for mode in ("sw", "nw"):
            row = []
            kern = "?"
            got = None
            for want_end in (False, True):
                b = ctx.batch(mode, seqs, pa, pb, 1, -1, -1, want_end)
                b.run()
                b.last_ms()
                t = []
                for _ in range(3):
                    b.run()
                    t.append(b.last_ms())
                if not want_end:
                    kern = b.info()["kernel"]
                    got = b.fetch(numpy_out=True)
                else:
                    g2 = b.fetch(numpy_out=True)[0]
                    assert (g2 == got).all(), "engines disagree on " + name
                b.close()
                row.append(min(t))
            # a sample against the oracle
            for k in list(range(min(n, 3))) + [n - 1]:
                want = O.score(mode, seqs[pa[k]], seqs[pb[k]], 1, -1, -1)[0]
                assert int(got[k]) == want, (name, mode, k, int(got[k]), want)
            print("%-52s %-6s %12.3f %12.3f  %s" % (name, mode, row[0], row[1], kern), flush=True)
    # hw2_amd -l on the C2 pair
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "p.fa"), "wb") as f:
            f.write(b">p\n" + bench.gen(1, 0, 0, 10000) + b"\n")
        with open(os.path.join(d, "t.fa"), "wb") as f:
            f.write(b">t\n" + bench.gen(1, 1, 0, 10000) + b"\n")
        for flag in ("-l", "-g"):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                subprocess.run([pkg.CLI_PATH, flag, "-p", os.path.join(d, "p.fa"), "-t", os.path.join(d, "t.fa"), "-o",
                                os.path.join(d, "o.txt"), "-s", "1", "-1", "-1"], check=True)
                best = min(best, time.perf_counter() - t0)
            dbg = subprocess.run([pkg.CLI_PATH, flag, "-p", os.path.join(d, "p.fa"), "-t", os.path.join(d, "t.fa"), "-o",
                                  os.path.join(d, "o.txt"), "-s", "1", "-1", "-1"], env=dict(os.environ, PWA_DEBUG="1"),
                                 capture_output=True, text=True).stderr
            dev = [l for l in dbg.splitlines() if "fill + walk" in l or "run + fetch" in l or "scores pass" in l]
            print("hw2_amd %s on the C2 pair: wall %.3f s (process start + HIP init included)  %s" % (flag, best, " | ".join(x.strip() for x in dev)))
            print(open(os.path.join(d, "o.txt")).read()[:120].replace("\n", " / "))
    ctx.close()


if __name__ == "__main__":
    main()
