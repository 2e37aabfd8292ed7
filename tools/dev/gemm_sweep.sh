#!/bin/bash
# per-shape timings of the 1x1-conv GEMM under different tile-dispatch settings (one process per setting: the knobs are read once)
out=gpurun_out/gemm_sweep; mkdir -p $out
run() { name=$1; shift; env "$@" GEMM_BENCH_BF16=$BF16 GEMM_BENCH_ALL=1 python tools/dev/gemm_bench.py 0 2>/dev/null | grep -E "^ALL|weighted" > $out/$name.txt; head -1 $out/$name.txt; }
run default X=1
run nosq MMD_SQ_TILES=0
run noskinny MMD_SKINNY_TILES=0
run allsq MMD_SQ_TILES=100000
run skinny400 MMD_SKINNY_TILES=400 MMD_SQ_MIN=400
run bn32 MMD_BN32_GAIN=0
python - <<'P'
import glob, os, collections
res = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/gemm_sweep/*.txt")):
    n = os.path.basename(f)[:-4]
    for line in open(f):
        if line.startswith("ALL"):
            _, M, K, N, fl, c, t = line.strip().split(",")
            res[(int(M), int(K), int(N), int(fl), int(c))][n] = float(t)
tot_d = tot_b = 0
rows = []
for k, v in res.items():
    if k[3] < 0: continue
    d = v["default"]; b = min(v.values()); bn = min(v, key=v.get)
    tot_d += d * k[4]; tot_b += b * k[4]
    rows.append(((d - b) * k[4], k, d, b, bn, v))
print("default %.3f ms  best-of %.3f ms" % (tot_d / 1e3, tot_b / 1e3))
for g, k, d, b, bn, v in sorted(rows, reverse=True)[:40]:
    print("gain %6.1f us/step  M%-7d K%-5d N%-5d f%-3d n=%3d  default %6.1f  best %6.1f (%s)  %s" % (g, *k, d, b, bn, {a: round(x, 1) for a, x in v.items()}))
P
