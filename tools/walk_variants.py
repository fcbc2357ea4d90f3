"""Walk-time A/B of libpwalign builds (PWA_LIB): the C5 pair's op-list walk, traceback_ms of the best of a few runs.
usage: python tools/walk_variants.py [n] [nw|sw]   (timing-only variants produce wrong op lists; nothing is verified here)"""
import importlib.util, os, sys
import torch  # noqa: F401  (first)
spec = importlib.util.spec_from_file_location("pwa_pkg", os.path.join(os.path.dirname(__file__), "..", "bioinformatics-algorithms_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from bench import gen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = pkg.Context(0)
a, b = gen(1, 0, 0, n), gen(1, 1, 0, n)
mode = sys.argv[2] if len(sys.argv) > 2 else "nw"
fills, walks = [], []
for _ in range(6):
    r = ctx.align(mode, a, b, 1, -1, -1, raw=True)
    st = ctx.align_stats()
    fills.append(st["fill_ms"]); walks.append(st["traceback_ms"])
print(os.path.basename(os.environ.get("PWA_LIB", "in-tree")), "free" if not os.environ.get("PWA_NO_HEAD_FREE") else "guard", mode, n, "n_ops", len(r["ops"]), "score", r["score"],
      "fill min %.3f median %.3f  walk min %.3f ms" % (min(fills), sorted(fills)[len(fills) // 2], min(walks)))
