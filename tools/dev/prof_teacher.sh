#!/bin/bash
# per-kernel breakdown of one frozen teacher forward (GPU box, repo root): gpurun_out/teacher/summary.txt
export TMPDIR=/tmp
out=gpurun_out/teacher; rm -rf $out; mkdir -p $out
N=10
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 tools/dev/prof_teacher.py $N > $out/trace.log 2>&1 || tail -5 $out/trace.log
grep "teacher forward" $out/trace.log
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$ks" $N > $out/summary.txt <<'PY'
import csv, sys
n = int(sys.argv[2]) + 2 + 1 + 23       # eager runs + warm-up + capture + graph replays all go through the trace
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel time per forward: {tot / n / 1e3:.1f} us over {sum(int(r['Calls']) for r in rows) / n:.0f} launches")
for r in rows[:45]:
    print(f"{r['Name'][:70]:<70} n={int(r['Calls']) / n:6.1f} us/fwd={float(r['TotalDurationNs']) / n / 1e3:8.1f} avg={float(r['AverageNs']) / 1e3:7.1f}")
PY
head -50 $out/summary.txt
rm -rf $out/trace
