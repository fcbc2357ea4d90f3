#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile.sh) into the markdown summary kept under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

out, workload, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def rows(pattern):
    fs = glob.glob(os.path.join(out, pattern), recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []


print("# rocprofv3 summary -- workload %s (%s)\n" % (workload, tag))
print("Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s --steps 3 --warmup 1 --no-cpu-baseline`" % workload)
print("(PMC passes: same command with `--pmc ...` only, one counter group per run.)\n")
bench_line = None
for l in open(os.path.join(out, "trace.log")):
    if l.startswith("{"):
        bench_line = json.loads(l)
if bench_line:
    print("bench line under the profiler: value = %.1f %s, ms_per_step = %.3f, kernel_ms = %s\n"
          % (bench_line["value"], bench_line["unit"], bench_line["ms_per_step"], bench_line["roofline"].get("kernel_ms")))
print("## kernel stats (--kernel-trace --stats)\n")
print("| kernel | calls | total ms | avg ms | min ms | max ms | % |")
print("|---|---|---|---|---|---|---|")
for r in rows("trace/**/*kernel_stats.csv"):
    print("| `%s` | %s | %.3f | %.4f | %.4f | %.4f | %s |" % (r["Name"], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
          float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
print("\n## counters (mean per dispatch of each kernel)\n")
print("| kernel | counter | dispatches | mean value |")
print("|---|---|---|---|")
meta = {}
for d in ("pmc_sq", "pmc_fetch", "pmc_write"):
    agg = collections.defaultdict(list)
    for r in rows(d + "/**/*counter_collection.csv"):
        if r["Kernel_Name"].startswith("__amd"):
            continue
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        meta[r["Kernel_Name"]] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                                  r["Grid_Size"], r["Workgroup_Size"])
    for (k, c), v in sorted(agg.items()):
        print("| `%s` | %s | %d | %.6g |" % (k, c, len(v), sum(v) / len(v)))
print("\n## dispatch resources\n")
print("| kernel | VGPR | AGPR | SGPR | LDS B | scratch B | grid | workgroup |")
print("|---|---|---|---|---|---|---|---|")
for k, m in meta.items():
    print("| `%s` | %s |" % (k, " | ".join(m)))
print("\nNotes: FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3); on gfx950 FETCH_SIZE reports half the bytes of wide "
      "coalesced reads (MI355X_MICROARCH.md, HBM) -- bench.py/DESIGN.md apply that correction where they quote traffic. "
      "GRBM_GUI_ACTIVE is summed over the 8 XCDs (divide by 8 for cycles).")
