"""HBM traffic of the 1x1-conv GEMM family per SHAPE: joins the per-dispatch FETCH_SIZE / WRITE_SIZE counters of the final eager step of a
`rocprofv3 --pmc` run of bench.py with the launch tags bench.py dumps for that same step (MMD_PROF_DUMP: family, tag, us, flops, algorithmic
bytes - one line per pw_dispatch, in launch order).  FETCH_SIZE doubled (gfx950 correction), KiB units (MI355X_MICROARCH.md, HBM section).
Usage: pmc_by_shape.py <fetch counter_collection.csv> <write counter_collection.csv> <prof dump csv>"""
import collections, csv, sys

KEYS = ("pw_gemm_kernel", "pw_gemm_skinny_kernel", "pw_stream_kernel", "pw_rows_kernel", "pw_longk_kernel", "mbconv_expand_bwd_kernel", "pw_slab_kernel")
# (a K-sliced slab launch is two kernels under ONE tag: the combine kernel's bytes are added to the slab kernel's in front of it)


def final_eager_step(rows):
    """rows (sorted by dispatch) behind the last optimizer launch = bench.py's eager single-stream step(s).  Since round 6 the bench runs that
    step twice (a warm one, then the one its HIP events bracket).  A kernel that runs exactly once per step marks the period: the last
    `period` launches are the final step (the warm step may differ by one-off launches, so the halves are not compared)."""
    ad = [i for i, r in enumerate(rows) if "adam2_kernel" in r["Kernel_Name"]]
    seg = rows[ad[-1] + 1:]
    for marker in ("focal_finalize_kernel", "mta_kl_multi_kernel"):
        pos = [i for i, r in enumerate(seg) if marker in r["Kernel_Name"]]
        if len(pos) >= 2:
            return seg[len(seg) - (pos[-1] - pos[-2]):]
        if len(pos) == 1:
            return seg
    return seg


def seq(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = []
    for r in final_eager_step(rows):
        n = r["Kernel_Name"]
        if "pw_slab_combine_kernel" in n and out:
            out[-1] = (out[-1][0], out[-1][1] + float(r["Counter_Value"]) * 1024.0)
        elif any(k in n for k in KEYS) and "false, 2>" not in n and ", 0, 2>" not in n:      # PRO = 2 is the stem's implicit GEMM: launched outside pw_dispatch, no tag
            out.append((n.split("(")[0], float(r["Counter_Value"]) * 1024.0))
    return out


f, w = seq(sys.argv[1]), seq(sys.argv[2])
tags = []
for line in open(sys.argv[3]):
    head, us, fl, by = line.rstrip("\n").rsplit(",", 3)
    fam, tag = head.split(",", 1)
    if fam.strip() == "0":
        tags.append((tag, float(by), float(us)))
if not (len(f) == len(w) == len(tags)):
    sys.exit("dispatch / tag count mismatch: fetch %d write %d tags %d" % (len(f), len(w), len(tags)))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0, ""])
for (kn, fb), (_, wb), (tag, alg, us) in zip(f, w, tags):
    a = agg[tag]
    a[0] += 1; a[1] += 2.0 * fb; a[2] += wb; a[3] += alg; a[4] += us; a[5] = kn[:44]
tot_m = sum(a[1] + a[2] for a in agg.values()); tot_a = sum(a[3] for a in agg.values())
print("1x1-conv GEMM family, one eager step: %d launches, measured %.2f GB (2 x FETCH_SIZE + WRITE_SIZE), algorithmic %.2f GB, ratio %.2f" % (
    len(tags), tot_m / 1e9, tot_a / 1e9, tot_m / tot_a))
print("flags f: 1 producer activation, 2 gate, 4 statistics, 8 residual, 16 folded BN, 32 pyramid; sorted by excess bytes")
print("%-34s %4s %9s %9s %9s %6s %8s  %s" % ("shape", "n", "fetch MB", "write MB", "alg MB", "ratio", "excess MB", "kernel"))
for tag, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2] - kv[1][3]))[:40]:
    m = a[1] + a[2]
    print("%-34s %4d %9.1f %9.1f %9.1f %6.2f %8.1f  %s" % (tag, a[0], a[1] / 1e6, a[2] / 1e6, a[3] / 1e6, m / max(a[3], 1), (m - a[3]) / 1e6, a[5]))
