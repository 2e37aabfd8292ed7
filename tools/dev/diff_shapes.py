"""Per-shape differences between two prof_by_shape listings: diff_shapes.py a.txt b.txt [families, default 0,1]"""
import re, sys
def load(f):
    d = {}
    for line in open(f):
        m = re.match(r'\s*(\d+) (.+?)\s+n=\s*([\d.]+) us/step=\s*([\d.]+) avg=\s*([\d.]+)us', line)
        if m: d[m.group(1) + ' ' + m.group(2).strip()] = (float(m.group(3)), float(m.group(4)), float(m.group(5)))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
fams = (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")
rows = sorted(((a[k][1] - b[k][1], k, a[k], b[k]) for k in a if k in b and k.split()[0] in fams), reverse=True)
for r in rows[:30]: print(f"{r[1]:45s} n={r[2][0]:.0f} {r[2][2]:7.1f} -> {r[3][2]:7.1f} us  ({r[0]:+.1f} us/step)")
print("...")
for r in rows[-12:]: print(f"{r[1]:45s} n={r[2][0]:.0f} {r[2][2]:7.1f} -> {r[3][2]:7.1f} us  ({r[0]:+.1f} us/step)")
print("sum", round(sum(r[0] for r in rows), 1), "us/step over", len(rows), "shapes;  totals", round(sum(r[2][1] for r in rows)), "->", round(sum(r[3][1] for r in rows)))
