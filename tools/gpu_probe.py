"""Step-by-step probe of the HIP path with progress lines (used when a GPU run misbehaves)."""
import faulthandler
import os
import sys
import time

faulthandler.dump_traceback_later(int(os.environ.get("PROBE_TIMEOUT", "50")), exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G  # noqa: E402
import oracle_lib as O  # noqa: E402


def say(*a):
    print("[%7.2f]" % (time.time() - T0), *a, flush=True)


T0 = time.time()
step = sys.argv[1] if len(sys.argv) > 1 else "all"
pkg = G.load_pkg()
say("lib loaded", pkg.lib().pwa_version())
ctx = pkg.Context(0)
say("context ok")
if step in ("k1", "all"):
    pats = [O.gen(1, 0, i, 150) for i in range(64)]
    txt = O.gen(1, 1, 0, 64)
    seqs = pats + [txt]
    say("k1 tiny: launching")
    got = ctx.scores("sw", seqs, list(range(64)), [64] * 64, 1, -1, -1)
    want = [O.score("sw", p, txt, 1, -1, -1)[0] for p in pats]
    say("k1 tiny SW", "OK" if got == want else ("MISMATCH", got[:8], want[:8]))
    got = ctx.scores("nw", seqs, list(range(64)), [64] * 64, 1, -1, -1)
    want = [O.score("nw", p, txt, 1, -1, -1)[0] for p in pats]
    say("k1 tiny NW", "OK" if got == want else ("MISMATCH", got[:8], want[:8]))
if step in ("k2", "all"):
    p, t = O.gen(1, 0, 0, 70), O.gen(1, 1, 0, 90)
    for mode in ("sw", "nw"):
        say("k2", mode, "launching")
        g = ctx.align(mode, p, t, 1, -1, -1, raw=True)
        w = O.align(mode, p, t, 1, -1, -1)
        ok = g["score"] == w["score"] and g["ops"] == w["ops"]
        say("k2", mode, "OK" if ok else ("MISMATCH", g["score"], w["score"], g["ops"][:40], w["ops"][:40], g["end"], w["end"]))
say("done")
