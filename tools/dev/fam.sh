#!/bin/bash
# per-family kernel time of the bench's roofline leg for the current environment: fam.sh [label]
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python tools/dev/fam.py "$1"
