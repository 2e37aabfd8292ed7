#!/bin/bash
# kernel-trace durations + SQ counters of the fused expand+depthwise kernel on the microbench (GPU box, repo root)
export TMPDIR=/tmp
out=gpurun_out/mbx; rm -rf $out; mkdir -p $out
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 tools/dev/one_mbx.py 10 > $out/trace.log 2>&1
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1)
head -12 $ks | cut -c1-160
set1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
set2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
i=0
for s in "$set1" "$set2"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $s --kernel-trace --output-format csv -d $out/pmc$i -o p -- python3 tools/dev/one_mbx.py 2 > $out/pmc$i.log 2>&1 || tail -3 $out/pmc$i.log
  f=$(find $out/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    if "mbx" in k or "dw_fwd" in k or "pw_" in k:
        print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
done
rm -rf $out/trace $out/pmc1 $out/pmc2
