#!/bin/bash
# bench.py with the 16-wave 3x3 kernel enabled from a minimum channel count upwards (9999 = never)
for c in 9999 512 256 128 9999 512; do
  out=$(ASL_WIDE_MIN_C=$c timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1) || exit 1
  echo "wide from C>=$c: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
