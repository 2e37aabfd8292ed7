"""Aggregate a MMD_PROF_DUMP csv (family,tag,us,flops,bytes) by tag: count, total us, TF/s, TB/s."""
import sys, collections
rows = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for line in open(sys.argv[1]):
    head, us, fl, by = line.rstrip("\n").rsplit(",", 3)
    fam, tag = head.split(",", 1)
    r = rows[(fam, tag)]
    r[0] += 1; r[1] += float(us); r[2] += float(fl); r[3] += float(by)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = sum(r[1] for r in rows.values())
print(f"total {tot / steps / 1e3:.3f} ms/step")
for (fam, tag), r in sorted(rows.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print(f"{fam:>2} {tag:<40} n={r[0] / steps:6.1f} us/step={r[1] / steps:8.1f} avg={r[1] / r[0]:7.1f}us "
          f"TF={r[2] / r[1] / 1e6:6.2f} TB/s={r[3] / r[1] / 1e6:5.2f}")
