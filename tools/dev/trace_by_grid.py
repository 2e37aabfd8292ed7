"""Per (kernel, grid size) launch statistics of one bench step from a rocprofv3 kernel trace: which SHAPES of a kernel cost the time
(the by-name summary hides that a kernel runs 8-block and 4096-block grids).  usage: trace_by_grid.py <kernel_trace.csv> [steps] [top]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam2_kernel" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
seg = rows[ad[-n - 2] + 1:ad[-2] + 1]
gk = "Grid_Size" if "Grid_Size" in seg[0] else "Grid_Size_X"
wk = "Workgroup_Size" if "Workgroup_Size" in seg[0] else "Workgroup_Size_X"
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    wg = max(int(r[wk]), 1)
    k = (r["Kernel_Name"].split("(")[0][:52], int(r[gk]) // wg)
    agg[k][0] += 1
    agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 120
print("per step: kernel, blocks, launches, total us, avg us")
for (k, g), v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-54s blocks=%7d n=%5.1f us=%8.1f avg_us=%7.1f" % (k, g, v[0] / n, v[1] / n / 1e3, v[1] / v[0] / 1e3))
