"""Walk-time A/B of libpwalign builds (PWA_LIB): the C5 pair's op-list walk, traceback_ms of the best of a few runs.
usage: python tools/walk_variants.py [n]   (timing-only variants produce wrong op lists; nothing is verified here)"""
import importlib.util, os, sys
import torch  # noqa: F401  (first)
spec = importlib.util.spec_from_file_location("pwa_pkg", os.path.join(os.path.dirname(__file__), "..", "bioinformatics-algorithms_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from bench import gen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = pkg.Context(0)
a, b = gen(1, 0, 0, n), gen(1, 1, 0, n)
best = None
for _ in range(4):
    r = ctx.align("nw", a, b, 1, -1, -1, raw=True)
    st = ctx.align_stats()
    if best is None or st["traceback_ms"] < best["traceback_ms"]: best = st
print(os.environ.get("PWA_LIB", "in-tree"), "n_ops", len(r["ops"]), "score", r["score"], "fill %.3f walk %.3f ms" % (best["fill_ms"], best["traceback_ms"]))
