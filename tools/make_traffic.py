#!/usr/bin/env python3
"""profiles/<tag>_<workload>_rocprof_summary.md -> profiles/traffic_<workload>.json: HBM bytes per launch of the dominant kernel
from the PMC passes, corrected as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE
counts wide coalesced reads at half their bytes (x2); WRITE_SIZE is exact for wide streaming stores."""
import json
import re
import sys

tag, w = sys.argv[1], sys.argv[2]
path = "profiles/%s_%s_rocprof_summary.md" % (tag, w)
txt = open(path).read()
kern = None
best = 0.0
for m in re.finditer(r"^\| `([^`]+)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", txt, re.M):   # kernel stats rows: name, calls, total ms, avg ms
    if float(m.group(3)) > best:
        best, kern, avg_ms = float(m.group(3)), m.group(1), float(m.group(4))
vals = {}
for m in re.finditer(r"^\| `([^`]+)` \| (\w+) \| (\d+) \| ([\d.e+]+) \|", txt, re.M):
    if m.group(1) == kern:
        vals[m.group(2)] = float(m.group(4))
fetch, write = vals.get("FETCH_SIZE", 0.0) * 1024, vals.get("WRITE_SIZE", 0.0) * 1024
out = {"kernel": kern, "kernel_avg_ms": avg_ms, "hbm_bytes_per_launch": 2 * fetch + write, "hbm_bytes_per_launch_uncorrected": fetch + write,
       "fetch_size_kib": vals.get("FETCH_SIZE"), "write_size_kib": vals.get("WRITE_SIZE"),
       "hbm_GBs": (2 * fetch + write) / (avg_ms * 1e-3) / 1e9,
       "source": "%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; mean per dispatch of %s)" % (path, kern)}
json.dump(out, open("profiles/traffic_%s.json" % w, "w"), indent=1)
print(json.dumps(out))
