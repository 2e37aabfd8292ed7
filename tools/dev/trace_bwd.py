"""Backward-only view of a rocprofv3 kernel trace of bench.py: kernels between focal_loss_kernel and adam2_kernel
   of the last full steps; per-kernel totals, idle gaps, and a coarse timeline (ms since segment start)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam2_kernel" in r["Kernel_Name"]]
fo = [i for i, r in enumerate(rows) if "focal_loss_kernel" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
segs = []
for a in ad[-n - 1:-1]:
    f = max(i for i in fo if i < a)
    segs.append(rows[f + 1:a])
agg = collections.defaultdict(lambda: [0, 0])
wall = gap = 0
for seg in segs:
    wall += int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    ce = None
    for r in seg:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if ce is not None and a > ce:
            gap += a - ce
        ce = b if ce is None else max(ce, b)
        k = r["Kernel_Name"].split("(")[0][:44]
        agg[k][0] += 1; agg[k][1] += b - a
print("backward: wall %.2f ms  sum-kernel %.2f ms  idle gaps %.2f ms  launches %d" % (
    wall / n / 1e6, sum(v[1] for v in agg.values()) / n / 1e6, gap / n / 1e6, sum(v[0] for v in agg.values()) / n))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-46s n=%5.0f ms=%7.3f avg_us=%6.1f" % (k, v[0] / n, v[1] / n / 1e6, v[1] / v[0] / 1e3))
# timeline of the last segment in 0.5 ms buckets: dominant kernel per bucket
seg = segs[-1]
t0 = int(seg[0]["Start_Timestamp"])
b = collections.defaultdict(lambda: collections.Counter())
for r in seg:
    a, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    b[a // 500000][r["Kernel_Name"].split("(")[0][:30]] += e - a
for k in sorted(b):
    tot = sum(b[k].values())
    print("%5.1f ms busy %3d%% n=%3d  %s" % (k * 0.5, tot / 5000, sum(1 for r in seg if (int(r["Start_Timestamp"]) - t0) // 500000 == k),
                                   ", ".join("%s %.0f" % (n_, v / 1e3) for n_, v in b[k].most_common(3))))
# the tail: the last launches in front of the optimizer (what the step's end waits for), times relative to the segment's end
t1 = int(seg[-1]["End_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
print("tail (us before the backward's end): start  end  queue  kernel")
for r in sorted(seg, key=lambda r: int(r["End_Timestamp"]))[-28:]:
    print("  %8.1f %8.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t1) / 1e3, (int(r["End_Timestamp"]) - t1) / 1e3,
                                      r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:60]))
if len(sys.argv) > 3:      # every launch of the last segment, in start order (times in us since the segment's start)
    with open(sys.argv[3], "w") as f:
        for r in seg:
            gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(int(r.get("Grid_Size_Y", 1) or 1), 1) * max(int(r.get("Grid_Size_Z", 1) or 1), 1)
            wx = max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1) * max(int(r.get("Workgroup_Size_Y", 1) or 1), 1) * max(int(r.get("Workgroup_Size_Z", 1) or 1), 1)
            f.write("%9.1f %9.1f %7.1f  q%-3s blocks %6d  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                     (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), gx // wx,
                                                     r["Kernel_Name"].split("(")[0][:70]))
if len(sys.argv) > 4:      # the whole last step (from the optimizer launch of the step before), same columns
    a1 = ad[-2]; a0 = ad[-3]
    st = rows[a0 + 1:a1 + 1]
    t0 = int(st[0]["Start_Timestamp"])
    with open(sys.argv[4], "w") as f:
        for r in st:
            gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(int(r.get("Grid_Size_Y", 1) or 1), 1) * max(int(r.get("Grid_Size_Z", 1) or 1), 1)
            wx = max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1) * max(int(r.get("Workgroup_Size_Y", 1) or 1), 1) * max(int(r.get("Workgroup_Size_Z", 1) or 1), 1)
            f.write("%9.1f %9.1f %7.1f  q%-3s blocks %6d  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                     (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), gx // wx,
                                                     r["Kernel_Name"].split("(")[0][:70]))
