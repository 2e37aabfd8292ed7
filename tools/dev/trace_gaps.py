"""Largest idle gaps (no kernel running on any queue) inside the last full steps of a rocprofv3 kernel trace of bench.py, with the kernels on
either side.  usage: trace_gaps.py <kernel_trace.csv> [steps] [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
ad = [i for i, r in enumerate(rows) if "adam2_kernel" in r["Kernel_Name"]]
for k in range(1, n + 1):
    seg = rows[ad[-k - 2] + 1:ad[-k - 1] + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    gaps, ce, prev = [], None, None
    for r in seg:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if ce is not None and a > ce:
            gaps.append((a - ce, (ce - t0) / 1e6, prev["Kernel_Name"].split("(")[0][:40], r["Kernel_Name"].split("(")[0][:40]))
        if ce is None or b > ce:
            ce, prev = b, r
    wall = (int(seg[-1]["End_Timestamp"]) - t0) / 1e6
    print("step -%d: wall %.3f ms, idle %.3f ms in %d gaps" % (k, wall, sum(g[0] for g in gaps) / 1e6, len(gaps)))
    for g in sorted(gaps, reverse=True)[:top]:
        print("   %7.1f us at %7.3f ms   %s -> %s" % (g[0] / 1e3, g[1], g[2], g[3]))
