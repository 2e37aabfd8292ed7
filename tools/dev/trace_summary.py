"""Summarise a rocprofv3 kernel trace of bench.py per training step (steps delimited by the Adam kernel)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam2_kernel" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
s, e = ad[-n - 2], ad[-2]
seg = rows[s + 1:e + 1]
wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / n / 1e6
busy, cs, ce = 0, None, None
for r in seg:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if ce is None or a > ce:
        if ce is not None:
            busy += ce - cs
        cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"].split("(")[0][:44]
    agg[k][0] += 1
    agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("wall ms/step %.2f  busy-union %.2f  sum-kernel %.2f  launches/step %.0f" % (
    wall, busy / n / 1e6, sum(v[1] for v in agg.values()) / n / 1e6, len(seg) / n))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-46s n=%5.0f ms=%7.3f avg_us=%6.1f" % (k, v[0] / n, v[1] / n / 1e6, v[1] / v[0] / 1e3))
