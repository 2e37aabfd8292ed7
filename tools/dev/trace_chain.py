"""Launch-by-launch view of ONE serialised bench step (MMD_SERIAL=1: every launch alone, in order) from a rocprofv3 kernel trace:
position, kernel, blocks, duration and the gap to the previous launch's end, averaged over the traced steps by position.
usage: trace_chain.py <kernel_trace.csv> [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam2_kernel" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
segs = [rows[ad[-k - 2] + 1:ad[-k - 1] + 1] for k in range(1, n + 1)]
L = len(segs[0])
assert all(len(s) == L for s in segs), [len(s) for s in segs]


def blocks(r):
    t = 1
    for ax in "XYZ":
        g, w = r.get("Grid_Size_" + ax), r.get("Workgroup_Size_" + ax)
        if g is not None:
            t *= max(int(g), 1) // max(int(w), 1)
    return t


tot_d = tot_g = 0.0
print("pos  kernel                                              blocks   dur_us  gap_us   t_end_ms")
t = 0.0
for i in range(L):
    d = sum(int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"]) for s in segs) / n / 1e3
    g = sum((int(s[i]["Start_Timestamp"]) - int(s[i - 1]["End_Timestamp"])) if i else 0 for s in segs) / n / 1e3
    t += d + g
    tot_d += d; tot_g += g
    print("%4d %-52s %7d %7.1f %7.1f %9.3f" % (i, segs[0][i]["Kernel_Name"].split("(")[0][:52], blocks(segs[0][i]), d, g, t / 1e3))
print("launches %d  kernel time %.3f ms  gaps %.3f ms" % (L, tot_d / 1e3, tot_g / 1e3))
