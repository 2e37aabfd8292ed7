"""Per-family kernel time of the bench's roofline leg from a bench JSON line on stdin: fam.py [label]"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "ms/step", d["ms_per_step"], "frac", r.get("frac"), "achieved", r.get("achieved"), "ms", r.get("kernel_ms"))
for f in r.get("families", []):
    print("   ", f)
