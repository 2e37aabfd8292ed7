#!/bin/bash
# SQ counters per kernel over the bench step (GPU box, repo root): VALU / MFMA / LDS busy and wait cycles -> gpurun_out/sq/summary.txt
export TMPDIR=/tmp
out=gpurun_out/sq; rm -rf $out; mkdir -p $out
set1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"
set2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS"
i=0
for s in "$set1" "$set2"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $s --kernel-trace --output-format csv -d $out/pmc$i -o p -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc$i.log 2>&1 || { tail -3 $out/pmc$i.log; exit 1; }
done
python3 - $out <<'PY' > $out/summary.txt
import csv, sys, collections, glob
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for i in (1, 2):
    f = glob.glob(f"{out}/pmc{i}/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ad = [j for j, r in enumerate(rows) if "adam2_kernel" in r["Kernel_Name"]]
    seg = rows[ad[-1] + 1:]          # the eager single-stream step(s) after the last optimizer launch
    # (round 6: the bench runs that step twice - a warm one first; a once-per-step kernel marks the period, the last `period` launches are the final step)
    # (one row per dispatch AND counter here: work in dispatch ids)
    mk = sorted({int(r["Dispatch_Id"]) for r in seg if "focal_finalize_kernel" in r["Kernel_Name"]})
    if len(mk) >= 2:
        last = int(seg[-1]["Dispatch_Id"])
        seg = [r for r in seg if int(r["Dispatch_Id"]) > last - (mk[-1] - mk[-2])]
    seen = set()
    for r in seg:
        k = r["Kernel_Name"].split("(")[0][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if i == 1 and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); calls[k] += 1
print("per kernel over one eager step: busy = SQ_BUSY_CYCLES/32 SEs (cycles); valu%, mfma%, lds% = pipe-busy share of busy*1024 SIMD-cycles; occ = average waves per SIMD; wait% of wave cycles")
tot = sum(v["SQ_BUSY_CYCLES"] for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"])[:45]:
    busy = v["SQ_BUSY_CYCLES"] / 32.0
    simd = busy * 1024 + 1
    print(f"{k:<48} n={calls[k]:4d} busy_us@2.1GHz={busy / 2100:8.1f} valu={4 * v['SQ_ACTIVE_INST_VALU'] / simd * 100:5.1f}% mfma={v['SQ_VALU_MFMA_BUSY_CYCLES'] / simd * 100:5.1f}% "
          f"lds={4 * v['SQ_ACTIVE_INST_LDS'] / simd * 100:5.1f}% occ={4 * v['SQ_WAVE_CYCLES'] / simd:4.1f} wait={v['SQ_WAIT_INST_ANY'] / (v['SQ_WAVE_CYCLES'] + 1) * 100:5.1f}% valu_inst/wave={v['SQ_INSTS_VALU'] / (v['SQ_WAVES'] + 1):7.0f} "
          f"mfma/wave={v['SQ_INSTS_MFMA'] / (v['SQ_WAVES'] + 1):6.0f} bankconf%={v['SQ_LDS_BANK_CONFLICT'] / (4 * v['SQ_ACTIVE_INST_LDS'] + 1) * 100:5.1f}")
PY
head -50 $out/summary.txt
rm -rf $out/pmc1 $out/pmc2
