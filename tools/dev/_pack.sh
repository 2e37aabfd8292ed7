for r in 1 2; do
MMD_DEV_NO_BWD=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/fwd+loss only, pack     /"
MMD_DEV_NO_BWD=1 MMD_NO_PACK=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -a "timed" | sed "s/^/fwd+loss only, no pack  /"
done
