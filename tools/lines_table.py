#!/usr/bin/env python3
"""bench lines (profiles/rNN_bench_lines.jsonl) -> the markdown rows of DESIGN.md section 6"""
import json
import sys

for l in open(sys.argv[1]):
    if not l.startswith("{"):
        continue
    j = json.loads(l)
    r = j["roofline"]
    w = j["config"]["workload"]
    cb = j.get("cpu_baseline") or {}
    extra = []
    if r.get("bound") == "hbm":
        alg = r.get("algorithmic_bytes_per_step") or r.get("algorithmic_bytes_per_launch")
        wr = r.get("written_bytes_per_step") or r.get("written_bytes_per_launch")
        extra.append("alg %.2f GB, written %.2f GB (x%.2f)" % (alg / 1e9, wr / 1e9, wr / alg))
        extra.append("walk %.3f ms" % r.get("traceback_ms", 0))
    else:
        extra.append("VALU %.2f iop/cell" % (r.get("valu_ops_per_cell") or 0))
        if r.get("issue_model"):
            extra.append("issue model ratio %.2f" % r["issue_model"]["ratio"])
    print("| %s | %.0f | %.3f | %.3f | %s %s | %s | %s |" % (
        w[:90], j["value"], j["ms_per_step"], r.get("kernel_ms") or 0, r.get("bound"), ("%.3f" % r["frac"]) if r.get("frac") else "-",
        "; ".join(extra), ("%.3f GCUPS %s x%d" % (cb.get("value", 0), cb.get("kind"), cb.get("cores", 1))) if cb else "-") + (" INVALID: " + j["invalid"] if j.get("invalid") else ""))
